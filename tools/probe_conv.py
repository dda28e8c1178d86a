import numpy as np, torch, sys
sys.path.insert(0, '.')
import oracle as orc, gpuaudiobench_amd as gab
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
T,B,L=2,512,4096
for k0 in (0,1,511,512,513,1000,3583,3584,4094,4095):
    ir=np.zeros((T,L),np.float32); ir[0,k0]=1.0; ir[1,(k0+7)%L]=0.5
    plan=gab.ConvPlan(T,B,L); plan.set_ir(dev(ir.ravel()))
    hist=np.zeros(T*L,np.float32); worst=0
    for n in range(11):
        x=orc.noise(T*B,seed=n+1)
        ref=orc.conv_accel_stream(x,ir.ravel(),hist,L,B,T,f64=True)
        y=plan.process(dev(x)).cpu().numpy()
        worst=max(worst,np.abs(y-ref).max())
    print("delta at",k0,"max abs err",worst)
    plan.close()
# real IR: error split
T=4
ir=orc.conv_accel_ir(L,T); plan=gab.ConvPlan(T,B,L); plan.set_ir(dev(ir))
hist=np.zeros(T*L,np.float32)
for n in range(12):
    x=orc.noise(T*B,seed=100+n)
    ref=orc.conv_accel_stream(x,ir,hist,L,B,T,f64=True)
    y=plan.process(dev(x)).cpu().numpy().astype(np.float64)
    e=np.abs(y-ref); i=e.argmax()
    print(n,"peak",np.abs(ref).max(),"maxerr",e.max(),"at s,t",divmod(i,T),"rms err",np.sqrt((e**2).mean()))
