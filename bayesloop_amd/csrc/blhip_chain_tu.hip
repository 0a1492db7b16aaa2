// One SLICE of the chain-resident kernels per compilation: -DBLC_TU=1 .. blcl::N_SLICES (build.py compiles the slices in parallel).
// Which instantiations a slice holds is decided here and nowhere else; blhip.hip dispatches on (tiles per wave, pass, ring length).
#include <algorithm>
#include "blhip_chain_launch.hpp"
#include "blhip_err.hpp"

#ifndef BLC_TU
#define BLC_TU 0        // (no slice selected -- a bare `hipcc -c` of this file: an empty object; build.py passes -DBLC_TU=1 .. N_SLICES)
#endif
#if BLC_TU >= 19
#include "blhip_chainax.hpp"      // (only the slices that hold its kernels: an edit of that header recompiles three units)
#endif

namespace {

using blerr::arm_kernel;
using blerr::fail;

// (ONE host function for all kernels of a signature -- the kernel arrives as a pointer --, the registry entry counted at the call site:
//  a launcher per instantiation was 2 KB of host code each, 3 MB of the library)
#define launch_chain_fn(K, s, Q, lds) (blreg::hit<&K>(), launch_chain_ptr(&K, s, Q, lds))
template <typename KernT>
void launch_chain_ptr(KernT KERN, hipStream_t s, const blc::ChainParams &Q, size_t lds) {
    arm_kernel(reinterpret_cast<const void *>(KERN));
    // (xch_mode bit 0 -- blc::chainax_kernel only: one XCD per chain, 8 x strips x ceil(chains / 8) blocks of which nslots x strips work)
    const unsigned blocks = (Q.xch_mode & 1) ? 8u * (unsigned)Q.strips * (unsigned)((Q.nslots + 7) / 8) : (unsigned)(Q.nslots * Q.strips);
    hipLaunchKernelGGL(KERN, dim3(blocks), dim3(blc::NT), lds, s, Q);
}

// the flavours of one (ring length, tiles per wave, direction).  <= 512 rows: forward and STORING backward passes, exact and padded (the
// folding backward pass is the two-chain kernel's); 1024 rows: forward, storing (exact) and FOLDING (exact, padded) backward passes
template <int NK, int NTW, bool BWD>
void launch_k(hipStream_t s, const blc::ChainParams &Q, bool store, bool pad) {
    const size_t lds = blc::lds_doubles<NK, NTW>() * sizeof(double);
    if constexpr (!BWD) {
        if (pad && store) launch_chain_fn((blc::chain_kernel<NK, NTW, false, true, true>), s, Q, lds);
        else if (pad) launch_chain_fn((blc::chain_kernel<NK, NTW, false, false, true>), s, Q, lds);
        else if (store) launch_chain_fn((blc::chain_kernel<NK, NTW, false, true, false>), s, Q, lds);
        else launch_chain_fn((blc::chain_kernel<NK, NTW, false, false, false>), s, Q, lds);
    } else if constexpr (NTW <= 4) {
        // (<= 512 rows: the folding backward pass is the two-chain kernel's -- blc::chain_fold2_kernel, which also takes an odd chain out;
        //  the one-chain folding flavour was reachable through a development option only and was pruned in round 6: 84 kernels)
        if (!store) fail("internal: chain-resident launch of a folding backward pass on <= 512 rows (the two-chain kernel folds there)");
        if (pad) launch_chain_fn((blc::chain_kernel<NK, NTW, true, true, true>), s, Q, lds);
        else launch_chain_fn((blc::chain_kernel<NK, NTW, true, true, false>), s, Q, lds);
    } else {
        if (pad && store) fail("internal: the 1024-row chain-resident kernels store posteriors on the exact geometry only");
        if (pad) launch_chain_fn((blc::chain_kernel<NK, NTW, true, false, true>), s, Q, lds);
        else if (store) launch_chain_fn((blc::chain_kernel<NK, NTW, true, true, false>), s, Q, lds);
        else launch_chain_fn((blc::chain_kernel<NK, NTW, true, false, false>), s, Q, lds);
    }
}

// tabulated likelihood: forward (stored / evidence-only), backward (stored / folded)
template <int NK, int NTW, bool PAD>
void launch_k_tab(hipStream_t s, const blc::ChainParams &Q, bool bwd, bool store) {
    const size_t lds = blc::lds_doubles<NK, NTW>() * sizeof(double);
    if constexpr (PAD) {                 // (padded grids: the folding backward pass would be the two-chain kernel's, which has no table flavour)
        if (bwd && !store) fail("internal: padded chain-resident launch of a folding backward pass (tabulated likelihood)");
        if (bwd) launch_chain_fn((blc::chain_kernel<NK, NTW, true, true, true, true>), s, Q, lds);
        else if (store) launch_chain_fn((blc::chain_kernel<NK, NTW, false, true, true, true>), s, Q, lds);
        else launch_chain_fn((blc::chain_kernel<NK, NTW, false, false, true, true>), s, Q, lds);
    } else {
        if (bwd && store) launch_chain_fn((blc::chain_kernel<NK, NTW, true, true, false, true>), s, Q, lds);
        else if (bwd) launch_chain_fn((blc::chain_kernel<NK, NTW, true, false, false, true>), s, Q, lds);
        else if (store) launch_chain_fn((blc::chain_kernel<NK, NTW, false, true, false, true>), s, Q, lds);
        else launch_chain_fn((blc::chain_kernel<NK, NTW, false, false, false, true>), s, Q, lds);
    }
}
template <int NTW, bool PAD>
void launch_w_tab(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store) {
    switch (nk) {
        case 4: launch_k_tab<4, NTW, PAD>(s, Q, bwd, store); break;
        case 6: launch_k_tab<6, NTW, PAD>(s, Q, bwd, store); break;
        case 8: launch_k_tab<8, NTW, PAD>(s, Q, bwd, store); break;
        case 10: launch_k_tab<10, NTW, PAD>(s, Q, bwd, store); break;
        case 12: launch_k_tab<12, NTW, PAD>(s, Q, bwd, store); break;
        case 14: launch_k_tab<14, NTW, PAD>(s, Q, bwd, store); break;
        case 16: launch_k_tab<16, NTW, PAD>(s, Q, bwd, store); break;
        case 18: launch_k_tab<18, NTW, PAD>(s, Q, bwd, store); break;
        case 20: launch_k_tab<20, NTW, PAD>(s, Q, bwd, store); break;
        case 22: launch_k_tab<22, NTW, PAD>(s, Q, bwd, store); break;
        case 24: launch_k_tab<24, NTW, PAD>(s, Q, bwd, store); break;
        default: fail("internal: chain-resident kernel (tabulated likelihood) with %d band blocks", nk);
    }
}

// ring lengths LO .. HI (even; 4 = the no-stencil kernels of change-point studies).  band = 16 + 2 R0 columns,
// R0 = 4, 8, ... 80 (NK = 6 .. 44; the slices hold NK <= 24 and NK >= 26 apart: the wide bands are the longer compilations)
#define BLC_CASE(NKV)                                                                                  \
    case NKV:                                                                                          \
        if constexpr (NKV >= LO && NKV <= HI) { launch_k<NKV, NTW, BWD>(s, Q, store, pad); return; }   \
        break;
template <int NTW, bool BWD, int LO, int HI>
void launch_w(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad) {
    switch (nk) {
        BLC_CASE(4) BLC_CASE(6) BLC_CASE(8) BLC_CASE(10) BLC_CASE(12) BLC_CASE(14) BLC_CASE(16) BLC_CASE(18) BLC_CASE(20) BLC_CASE(22) BLC_CASE(24)
        BLC_CASE(26) BLC_CASE(28) BLC_CASE(30) BLC_CASE(32) BLC_CASE(34) BLC_CASE(36) BLC_CASE(38) BLC_CASE(40) BLC_CASE(42) BLC_CASE(44)
        default: break;
    }
    fail("internal: chain-resident kernel with %d band blocks, %d tiles per wave", nk, NTW);
}
#undef BLC_CASE

template <int NK, int NTW>
void launch_fold2_k(hipStream_t s, const blc::ChainParams &Q, bool pad) {
    const size_t lds = blc::lds_doubles_fold2<NK, NTW>() * sizeof(double);
    const dim3 grid((unsigned)(((Q.nslots + 1) / 2) * Q.strips));
    if (pad) {
        arm_kernel(reinterpret_cast<const void *>(&blc::chain_fold2_kernel<NK, NTW, true>));
        BL_LAUNCH((blc::chain_fold2_kernel<NK, NTW, true>), grid, dim3(blc::NT), lds, s, Q);
    } else {
        arm_kernel(reinterpret_cast<const void *>(&blc::chain_fold2_kernel<NK, NTW, false>));
        BL_LAUNCH((blc::chain_fold2_kernel<NK, NTW, false>), grid, dim3(blc::NT), lds, s, Q);
    }
}

#define BLC_CASE(NKV)                                                                       \
    case NKV:                                                                               \
        if constexpr (NKV >= LO && NKV <= HI) { launch_fold2_k<NKV, NTW>(s, Q, pad); return; }   \
        break;
template <int NTW, int LO, int HI>
void launch_fold2_w(hipStream_t s, const blc::ChainParams &Q, int nk, bool pad) {
    switch (nk) {
        BLC_CASE(4) BLC_CASE(6) BLC_CASE(8) BLC_CASE(10) BLC_CASE(12) BLC_CASE(14) BLC_CASE(16) BLC_CASE(18) BLC_CASE(20) BLC_CASE(22) BLC_CASE(24)
        BLC_CASE(26) BLC_CASE(28) BLC_CASE(30) BLC_CASE(32) BLC_CASE(34) BLC_CASE(36) BLC_CASE(38) BLC_CASE(40) BLC_CASE(42) BLC_CASE(44)
        default: break;
    }
    fail("internal: two-chain fold kernel with %d band blocks, %d tiles per wave", nk, NTW);
}
#undef BLC_CASE

#if BLC_TU >= 19
// walks on both parameters: forward (stored / evidence-only), backward (stored / folded); ring lengths 8 .. 24 in steps of 4
template <int NK, int NTW, bool PAD>
void launch_k_ax(hipStream_t s, const blc::ChainParams &Q, bool bwd, bool store) {
    const size_t lds = blc::lds_doubles_ax<NK, NTW>() * sizeof(double);
    if (bwd && store) launch_chain_fn((blc::chainax_kernel<NK, NTW, true, true, PAD>), s, Q, lds);
    else if (bwd) launch_chain_fn((blc::chainax_kernel<NK, NTW, true, false, PAD>), s, Q, lds);
    else if (store) launch_chain_fn((blc::chainax_kernel<NK, NTW, false, true, PAD>), s, Q, lds);
    else launch_chain_fn((blc::chainax_kernel<NK, NTW, false, false, PAD>), s, Q, lds);
}
template <int NTW, bool PAD>
void launch_w_ax(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store) {
    switch (nk) {
        case 8: launch_k_ax<8, NTW, PAD>(s, Q, bwd, store); break;
        case 12: launch_k_ax<12, NTW, PAD>(s, Q, bwd, store); break;
        case 16: launch_k_ax<16, NTW, PAD>(s, Q, bwd, store); break;
        case 20: launch_k_ax<20, NTW, PAD>(s, Q, bwd, store); break;
        case 24: launch_k_ax<24, NTW, PAD>(s, Q, bwd, store); break;
        default: fail("internal: both-axes chain-resident kernel with %d band blocks", nk);
    }
}
#endif

}   // namespace

namespace blcl {

#if BLC_TU == 0
#elif BLC_TU == 1
void chain_ntw1(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store, bool pad) {
    if (bwd) launch_w<1, true, 4, 24>(s, Q, nk, store, pad); else launch_w<1, false, 4, 24>(s, Q, nk, store, pad);
}
void chain_ntw2(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store, bool pad) {
    if (bwd) launch_w<2, true, 4, 24>(s, Q, nk, store, pad); else launch_w<2, false, 4, 24>(s, Q, nk, store, pad);
}
#elif BLC_TU == 2
void chain_ntw3(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store, bool pad) {
    if (bwd) launch_w<3, true, 4, 24>(s, Q, nk, store, pad); else launch_w<3, false, 4, 24>(s, Q, nk, store, pad);
}
#elif BLC_TU == 3
void chain_ntw4_fwd(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad) { launch_w<4, false, 4, 24>(s, Q, nk, store, pad); }
#elif BLC_TU == 4
void chain_ntw4_bwd(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad) { launch_w<4, true, 4, 24>(s, Q, nk, store, pad); }
void fold2_ntw12(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool pad) {
    if (ntw == 2) launch_fold2_w<2, 4, 24>(s, Q, nk, pad);
    else if (ntw == 1) launch_fold2_w<1, 4, 24>(s, Q, nk, pad);
    else fail("internal: two-chain fold kernel with %d tiles per wave", ntw);
}
#elif BLC_TU == 5
void chain_ntw8_fwd_narrow(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad) { launch_w<8, false, 4, 24>(s, Q, nk, store, pad); }
#elif BLC_TU == 6
void chain_ntw8_bwd_narrow(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad) { launch_w<8, true, 4, 24>(s, Q, nk, store, pad); }
#elif BLC_TU == 7
void chain_ntw8_fwd_wide(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad) { launch_w<8, false, 26, 44>(s, Q, nk, store, pad); }
#elif BLC_TU == 8
void chain_ntw8_bwd_wide(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad) { launch_w<8, true, 26, 44>(s, Q, nk, store, pad); }
#elif BLC_TU == 9
void fold2_ntw34(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool pad) {
    if (ntw == 4) launch_fold2_w<4, 4, 24>(s, Q, nk, pad);
    else if (ntw == 3) launch_fold2_w<3, 4, 24>(s, Q, nk, pad);
    else fail("internal: two-chain fold kernel with %d tiles per wave", ntw);
}
#elif BLC_TU == 10
void chain_ntw12_wide(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool bwd, bool store, bool pad) {
    if (ntw == 2) { if (bwd) launch_w<2, true, 26, 44>(s, Q, nk, store, pad); else launch_w<2, false, 26, 44>(s, Q, nk, store, pad); }
    else if (ntw == 1) { if (bwd) launch_w<1, true, 26, 44>(s, Q, nk, store, pad); else launch_w<1, false, 26, 44>(s, Q, nk, store, pad); }
    else fail("internal: chain-resident kernel with %d tiles per wave", ntw);
}
#elif BLC_TU == 11
void chain_ntw3_wide(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store, bool pad) {
    if (bwd) launch_w<3, true, 26, 44>(s, Q, nk, store, pad); else launch_w<3, false, 26, 44>(s, Q, nk, store, pad);
}
#elif BLC_TU == 12
void chain_ntw4_fwd_wide(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad) { launch_w<4, false, 26, 44>(s, Q, nk, store, pad); }
#elif BLC_TU == 13
void chain_ntw4_bwd_wide(hipStream_t s, const blc::ChainParams &Q, int nk, bool store, bool pad) { launch_w<4, true, 26, 44>(s, Q, nk, store, pad); }
#elif BLC_TU == 14
void fold2_ntw12_wide(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool pad) {
    if (ntw == 2) launch_fold2_w<2, 26, 44>(s, Q, nk, pad);
    else if (ntw == 1) launch_fold2_w<1, 26, 44>(s, Q, nk, pad);
    else fail("internal: two-chain fold kernel with %d tiles per wave", ntw);
}
#elif BLC_TU == 15
void fold2_ntw34_wide(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool pad) {
    if (ntw == 4) launch_fold2_w<4, 26, 44>(s, Q, nk, pad);
    else if (ntw == 3) launch_fold2_w<3, 26, 44>(s, Q, nk, pad);
    else fail("internal: two-chain fold kernel with %d tiles per wave", ntw);
}
#elif BLC_TU == 16
void chain_ntw12_tab(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool bwd, bool store) {
    if (ntw == 2) launch_w_tab<2, false>(s, Q, nk, bwd, store);
    else if (ntw == 1) launch_w_tab<1, false>(s, Q, nk, bwd, store);
    else fail("internal: chain-resident kernel with %d tiles per wave", ntw);
}
#elif BLC_TU == 17
void chain_ntw34_tab(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool bwd, bool store) {
    if (ntw == 4) launch_w_tab<4, false>(s, Q, nk, bwd, store);
    else if (ntw == 3) launch_w_tab<3, false>(s, Q, nk, bwd, store);
    else fail("internal: chain-resident kernel with %d tiles per wave", ntw);
}
#elif BLC_TU == 18
void chain_ntw12_tab_pad(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool bwd, bool store) {
    if (ntw == 2) launch_w_tab<2, true>(s, Q, nk, bwd, store);
    else if (ntw == 1) launch_w_tab<1, true>(s, Q, nk, bwd, store);
    else fail("internal: chain-resident kernel with %d tiles per wave", ntw);
}
#elif BLC_TU == 19
void chainax_lik_transpose(hipStream_t s, const double *lik, double *out, int n0p, int n0t, int n1t, int T) {
    const long long G = (long long)n0p * n0p;
    BL_LAUNCH(blc::ax_lik_transpose_kernel, dim3((unsigned)std::min<long long>((G + 255) / 256, 1024), (unsigned)((T + 1) / 2)), dim3(256), 0, s, lik, out, n0p, n0t, n1t);
}
void chainax_lik_table(hipStream_t s, int n0p, int n0t, int n1t, int T, int d, int rec_len, const double *m0, const double *colA, const double *colB, const double *rec, double *out) {
    blc::AxLikParams L{n0p, n0t, n1t, T, d, rec_len, m0, colA, colB, rec, out};
    const long long G = (long long)n0p * n0p;
    BL_LAUNCH(blc::ax_lik_table_kernel, dim3((unsigned)std::min<long long>((G + 255) / 256, 1024), (unsigned)((T + 1) / 2)), dim3(256), 0, s, L);
}
#elif BLC_TU == 20
void chainax_ntw4(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store) { launch_w_ax<4, false>(s, Q, nk, bwd, store); }
#elif BLC_TU == 21
// (128 / 256 rows = columns: the PAD kernels also take the grids that fill their geometry -- no exact variants: 40 kernels less)
void chainax_ntw12_pad(hipStream_t s, const blc::ChainParams &Q, int nk, int ntw, bool bwd, bool store) {
    if (ntw == 2) launch_w_ax<2, true>(s, Q, nk, bwd, store);
    else if (ntw == 1) launch_w_ax<1, true>(s, Q, nk, bwd, store);
    else fail("internal: both-axes chain-resident kernel with %d tiles per wave", ntw);
}
#elif BLC_TU == 22
void chainax_ntw4_pad(hipStream_t s, const blc::ChainParams &Q, int nk, bool bwd, bool store) { launch_w_ax<4, true>(s, Q, nk, bwd, store); }
#else
#error "BLC_TU out of range"
#endif

}   // namespace blcl
