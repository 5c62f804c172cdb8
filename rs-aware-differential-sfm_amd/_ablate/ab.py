import sys, os, time, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import numpy as np, torch, rsdsfm
dev=torch.device('cuda',0)
data=rsdsfm.synth.make_config(2)
n=len(data['q']); t=data['truth']; v=t['v']/np.linalg.norm(t['v']); w=t['w']
sets=[dict(q=torch.from_numpy(data['q']).to(dev),u=torch.from_numpy(data['u']).to(dev),a=torch.from_numpy(data['alpha']).to(dev),ak=torch.from_numpy(data['alpha_k']).to(dev),rho=torch.empty(n,dtype=torch.float64,device=dev)) for _ in range(7)]
st=torch.cuda.Stream(dev); torch.cuda.set_stream(st)
s=rsdsfm.Solver(0,stream=st.cuda_stream)
def run(first,reps=200):
    for i in range(20):
        b=sets[i%7]; s.depth_lm_launch_dev(b['q'].data_ptr(),b['u'].data_ptr(),n,v,w,0.0,b['a'].data_ptr(),b['ak'].data_ptr(),b['rho'].data_ptr(),launch_id=0)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(reps):
        b=sets[i%7]; s.depth_lm_launch_dev(b['q'].data_ptr(),b['u'].data_ptr(),n,v,w,0.0,b['a'].data_ptr(),b['ak'].data_ptr(),b['rho'].data_ptr(),launch_id=0)
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e6
print(os.environ.get('RSDSFM_LIB','full')[-12:], 'first=1: %.1f us/launch'%run(True))
