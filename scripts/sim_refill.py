"""Design-time (CPU): wave iterations if free lanes took the next rays of a queue (/tmp/ray_visits.npy).  The model
said 12.0 -> 9.1 iterations per 64 rays at a threshold of 16 free lanes; the kernels built from it (k_shadow_rays writing
the rays to a queue, k_shadow_stream drawing from it) measured 360 ms for the walk against 348 ms for k_shadow with its
ray set-up included, for every threshold from 8 to 64, plus 60 ms for the kernel that writes the queue: rejected (the walk's
iteration count is set by the leaf-queue flushes as much as by the slowest lane)."""
import numpy as np
o=np.load("/tmp/ray_visits.npy")
order=np.lexsort((o[:,1],o[:,0])); o=o[order]
v=o[:,3:19]
rays=v.reshape(-1)          # device order: pair-major, rep inner (lane = pair*16+rep)
rays=np.where(rays<0,0,rays)  # no-ray lanes: zero work (they still occupy a lane slot in today's scheme)
NW=512
n=len(rays)//(NW*64)*(NW*64)
rays=rays[:n]
per_wave=n//NW
q=rays.reshape(NW,per_wave)   # each wave owns a contiguous chunk of the queue
# today
today=q.reshape(NW,-1,64).max(2).sum()
print("today: wave-iterations per 64 rays %.2f"%(today/(n/64)))
def sim(T, cost):
    pos=np.zeros(NW,int); rem=np.zeros((NW,64),int); iters=0.0; refills=0
    while True:
        idle=rem<=0
        nid=idle.sum(1)
        can=(pos<per_wave)
        do=can&((nid>=T)|(nid==64))
        if do.any():
            for w in np.nonzero(do)[0]:
                k=min(nid[w],per_wave-pos[w])
                lanes=np.nonzero(idle[w])[0][:k]
                rem[w,lanes]=q[w,pos[w]:pos[w]+k]; pos[w]+=k
            refills+=do.sum()
        active=(rem>0).any(1)
        if not active.any() and not (pos<per_wave).any(): break
        iters+=active.sum()
        rem=np.maximum(rem-1,0)
    return (iters+refills*cost)/(n/64), refills/(n/64)
for T in (64,32,16,8,4):
    for cost in (0.2,1.4):
        it,rf=sim(T,cost)
        print("refill when >= %2d lanes idle, refill cost %.1f iterations: %.2f wave-iterations per 64 rays (%.2f refills)"%(T,cost,it,rf))
