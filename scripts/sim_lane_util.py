"""Design-time (CPU): how evenly do the 64 walks of a k_shadow wavefront end?  From /tmp/ray_visits.npy
(sim_ray_visits.py).  Result: 7.9 visits per lane on average, 12.0 for the slowest lane (utilisation 0.66); the spread
is WITHIN a (hit, light) pair (a few repetitions that graze the mesh), so sorting or binning pairs by cost gains
nothing, and several repetitions of one pair per lane gain nothing either."""
import numpy as np
o=np.load("/tmp/ray_visits.npy")
# light-major queue, pixel order within a light (device k_pair_compact order)
order=np.lexsort((o[:,1],o[:,0])); o=o[order]
v=o[:,3:19].astype(float); v[v<0]=0   # reps outside the cone: no ray (lane idles)
full=o[:,2]==16
print("pairs",len(o),"rays/pair",o[:,2].mean(),"visits/ray",v.sum()/o[:,2].sum())
def waves(v, pairs_per_wave):
    n=len(v)//pairs_per_wave*pairs_per_wave
    return v[:n].reshape(-1,pairs_per_wave,v.shape[1])
# today: 4 pairs x 16 lanes, 1 rep per lane
w=waves(v,4); lane=w.reshape(len(w),-1)
print("today: mean over lanes %.2f, wave max %.2f, util %.3f"%(lane.mean(),lane.max(1).mean(),lane.mean()/lane.max(1).mean()))
# between-pair vs within-pair
pm=v.mean(1); print("std of per-ray visits %.2f; std of pair means %.2f; mean within-pair std %.2f"%(v.std(),pm.std(),v.std(1).mean()))
# R reps per lane, sequential: 16/R lanes per pair, 4R pairs per wave
for R in (2,4,8,16):
    w=waves(v,4*R)                       # (waves, pairs, 16)
    lane=w.reshape(len(w),4*R,16//R,R).sum(3).reshape(len(w),-1)   # lane sums over R consecutive reps
    print("R=%d: lane mean %.2f wave max %.2f util %.3f (iterations per ray %.3f vs today %.3f)"%(R,lane.mean(),lane.max(1).mean(),lane.mean()/lane.max(1).mean(), lane.max(1).mean()/R, waves(v,4).reshape(-1,64).max(1).mean()))
# R rays per lane taken from DIFFERENT pairs (lane handles rep k of pairs p, p+4, ...)
for R in (2,4):
    w=waves(v,4*R)
    lane=w.reshape(len(w),R,4,16).sum(1).reshape(len(w),-1)
    print("R=%d across pairs: util %.3f iterations per ray %.3f"%(R,lane.mean()/lane.max(1).mean(), lane.max(1).mean()/R))
print("--- sorted pairs (upper bound of what binning by a pilot repetition could give)")
for key,name in ((v.mean(1),"by pair mean (oracle)"),(v[:,0],"by rep 0 visits")):
    for light in (None,):
        idx=np.lexsort((key,o[:,0]))
        vs=v[idx]
        lane=waves(vs,4).reshape(-1,64)
        print(name,": util %.3f, iterations per ray %.3f"%(lane.mean()/lane.max(1).mean(), lane.max(1).mean()))
# binned by rep 0 into coarse bins, pixel order kept within a bin (coherence)
for nb in (4,8,16):
    b=np.minimum((v[:,0]/ (40/nb)).astype(int), nb-1)
    idx=np.lexsort((o[:,1],b,o[:,0]))
    lane=waves(v[idx],4).reshape(-1,64)
    print("bins",nb,": util %.3f, iterations per ray %.3f"%(lane.mean()/lane.max(1).mean(), lane.max(1).mean()))
