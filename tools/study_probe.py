import sys, time, numpy as np
sys.path.insert(0,'.')
import bayesloop_amd as bl, bench
eng=bl.get_engine()
for n,r in ((200,20),(128,20),(256,20),(300,30),(500,20)):
    T=1000
    d0=16.0/(n-1); d1=4.0/(n+1)
    for mode in (1,0):
        eng.set_option('chain_ax1',mode)
        S=bl.Study(silent=True); S.loadData(bench.series(3,T),silent=True)
        S.set(bl.om.Gaussian('mean',bl.cint(-8,8,n),'std',bl.oint(0,4,n)),
              bl.tm.CombinedTransitionModel(bl.tm.GaussianRandomWalk('s1',r/4.0*d0,target='mean'),bl.tm.GaussianRandomWalk('s2',r/4.0*d1,target='std')),silent=True)
        S.fit(silent=True, evidenceOnly=True)
        t0=time.perf_counter(); S.fit(silent=True, evidenceOnly=True); eng.synchronize(); dt=time.perf_counter()-t0
        tm=S.lastTiming
        t0=time.perf_counter(); S.fit(silent=True); eng.synchronize(); dt2=time.perf_counter()-t0
        tm2=S.lastTiming
        print('n %d radius %d chain_ax1 %d: evidence-only %.2f ms (variant %d, %.2f us per step); full fit %.2f ms (variants %d/%d: %.2f / %.2f us per step) logE %.10f'%(n,r,mode,dt*1e3,tm['fwd_kernel_variant'],tm['forward_ms']*1e3/T,dt2*1e3,tm2['fwd_kernel_variant'],tm2['bwd_kernel_variant'],tm2['forward_ms']*1e3/T,tm2['backward_ms']*1e3/T,S.logEvidence))
eng.set_option('chain_ax1',1)
