"""CPU diagnosis of the final-13682 lock-step state of LM iteration 6: which observations make the
float32 cost and increment of that state uncertain. numpy restatement of the residual (rotation matrix from the stored
quaternion, R p + t, Snavely projection - device_utils.hpp / oracle linearize_point) in float32 and float64 at the fixture
state; prints the cost difference, the observations and landmarks that carry it, and the per-camera share of the float32
oracle's increment error (fixture inc32 against the float64 referee). Needs tests/golden/_big/lockstep_final-13682_f32.npz.
Output kept in profiles/r6_final13682_iteration6_diagnosis.txt."""
import sys, types, numpy as np, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import bench
args = types.SimpleNamespace(translation_sigma=0.01, point_sigma=0.01, rotation_sigma=0.0)
prob = bench.make_problem("final-13682", args)[0]
z = np.load('/root/repo/tests/golden/_big/lockstep_final-13682_f32.npz')
cams = z['cams_6']; lms = z['lms_6']
ci = prob.obs_cam_idx; off = prob.lm_obs_offsets
li = np.repeat(np.arange(prob.n_lms), np.diff(off))
obs = prob.obs_xy
def resid(T, fma=False):
    c = cams.astype(T); p = lms.astype(T)[li]; o = obs.astype(T)
    q = c[ci,0:4]; t = c[ci,4:7]; f = c[ci,7]; k1=c[ci,8]; k2=c[ci,9]
    x,y,z,w = q[:,0],q[:,1],q[:,2],q[:,3]
    two=T(2); one=T(1)
    tx,ty,tz = two*x,two*y,two*z
    twx,twy,twz = tx*w,ty*w,tz*w
    txx,txy,txz = tx*x,ty*x,tz*x
    tyy,tyz,tzz = ty*y,tz*y,tz*z
    R0=one-(tyy+tzz);R1=txy-twz;R2=txz+twy;R3=txy+twz;R4=one-(txx+tzz);R5=tyz-twx;R6=txz-twy;R7=tyz+twx;R8=one-(txx+tyy)
    px = R0*p[:,0]+R1*p[:,1]+R2*p[:,2]+t[:,0]
    py = R3*p[:,0]+R4*p[:,1]+R5*p[:,2]+t[:,1]
    pz = R6*p[:,0]+R7*p[:,1]+R8*p[:,2]+t[:,2]
    mx=px/pz; my=py/pz; r2=mx*mx+my*my; rp=one+k1*r2+k2*r2*r2
    rx=f*mx*rp-o[:,0]; ry=f*my*rp-o[:,1]
    return rx,ry,pz
def cost(rx,ry):
    rs=(rx.astype(np.float64)**2+ry.astype(np.float64)**2)
    w=np.where(rs<1,1.0,1/np.sqrt(np.maximum(rs,1e-300)))
    return 0.5*(2-w)*w*rs
r64=resid(np.float64); r32=resid(np.float32)
c64=cost(r64[0],r64[1]); c32=cost(r32[0],r32[1])
print('cost64',c64.sum(),'cost32',c32.sum(), 'diff', c32.sum()-c64.sum())
d=np.abs(c32-c64); idx=np.argsort(-d)[:20]
print('sum abs diff', d.sum(), 'top20', d[idx].sum())
for i in idx[:12]:
    print(i, li[i], ci[i], 'c64',c64[i],'c32',c32[i],'pz64',r64[2][i],'pz32',r32[2][i],'|p|',np.linalg.norm(lms[li[i]]), 'k', off[li[i]+1]-off[li[i]])
lmn=np.linalg.norm(lms.astype(np.float64),axis=1)
print('landmark norms pct', np.percentile(lmn,[50,90,99,99.9,99.99,100]))
print('pz pct', np.percentile(r64[2],[0,0.001,0.01,0.1,1,50]))
# per landmark error mass
lmerr=np.bincount(li,weights=d,minlength=prob.n_lms); j=np.argsort(-lmerr)[:10]
print('top landmarks by cost err', [(int(a), float(lmerr[a]), float(lmn[a]), int(off[a+1]-off[a])) for a in j])

r = np.load('/root/repo/tests/golden/referee64_final-13682.npz')
for it in (3, 4, 5, 6, 7):
    a = z[f'inc32_{it}'].astype(np.float64).reshape(-1, 9)
    b = r[f'inc64_{it}'].reshape(-1, 9)
    d = np.linalg.norm(a - b, axis=1) ** 2
    tot = d.sum(); j = np.argsort(-d)[:4]
    print('iteration', it, 'float32 oracle vs float64 referee', np.sqrt(tot) / (np.linalg.norm(a) + np.linalg.norm(b)),
          'share of the squared error per camera:', [(int(c), round(float(d[c] / tot), 3)) for c in j])
