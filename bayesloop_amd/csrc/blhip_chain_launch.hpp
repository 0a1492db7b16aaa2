// Launchers of the chain-resident kernels (blhip_chainres.hpp).  The kernels are a few hundred template instantiations -- ring length NK x
// tiles per wave NTW x pass flavour -- and most of the library's compile time, so they are compiled as SLICES of blhip_chain_tu.hip in
// parallel (build.py: one object per BLC_TU value) and linked beside blhip.hip, which only sees these declarations.
#pragma once
#include <hip/hip_runtime.h>

#include "blhip_chainres.hpp"

namespace blcl {

// one chain per block: forward (store: every step's state is kept) / backward (store: posteriors kept; else folded into the partial accumulators)
void chain_ntw1(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store, bool pad);
void chain_ntw2(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store, bool pad);
void chain_ntw3(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store, bool pad);
void chain_ntw4_fwd(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad);
void chain_ntw4_bwd(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad);
void chain_ntw8_fwd_narrow(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad);      // NK <= 24
void chain_ntw8_bwd_narrow(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad);
void chain_ntw8_fwd_wide(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad);        // NK = 26 .. 44
void chain_ntw8_bwd_wide(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad);
// backward pass with the fused fold, two chains per block (blc::chain_fold2_kernel)
void fold2_ntw12(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool pad);
void fold2_ntw34(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool pad);
// ... the same shapes with the wide bands (radius 41 .. 80, NK = 26 .. 44)
void chain_ntw12_wide(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool bwd, bool store, bool pad);
void chain_ntw3_wide(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store, bool pad);
void chain_ntw4_fwd_wide(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad);
void chain_ntw4_bwd_wide(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad);
void fold2_ntw12_wide(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool pad);
void fold2_ntw34_wide(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool pad);

// ... and with the likelihood out of a table (blc::chain_kernel TAB: geometries of <= 512 rows, radius <= 40)
void chain_ntw12_tab(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool bwd, bool store);
void chain_ntw34_tab(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool bwd, bool store);
void chain_ntw12_tab_pad(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool bwd, bool store);

// ... walks on both parameters (blc::chainax_kernel: square geometries of 128 / 256 / 512 rows and columns, grids of any size inside them; band blocks 8 / 12 / 16 / 20 / 24
// = radius <= 8 .. 40 on either axis)
// (the likelihood of the even time steps in the transposed layout: out = ceil(T / 2) x n0p^2 doubles)
void chainax_lik_transpose(hipStream_t s, const double *lik, double *out, int n0p, int n0t, int n1t, int T);      // ... of a tabulated model
void chainax_lik_table(hipStream_t s, int n0p, int n0t, int n1t, int T, int d, int rec_len, const double *m0, const double *colA, const double *colB, const double *rec, double *out);
void chainax_ntw4(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store);
void chainax_ntw12_pad(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool bwd, bool store);      // grids smaller than the square geometry
void chainax_ntw4_pad(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store);

constexpr int N_SLICES = 22;      // BLC_TU = 1 .. N_SLICES (blhip_chain_tu.hip)

}   // namespace blcl
