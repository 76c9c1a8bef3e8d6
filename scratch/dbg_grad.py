import sys, numpy as np
sys.path.insert(0, '.')
from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import stack_models, synthetic_audio, synthetic_gmm_system
from oracle import oracle as O
ubm, spk = synthetic_gmm_system(n_speakers=3, C=256, D=72)
models=[ubm]+spk
e=Engine(0); e.load_gmm(models); e.set_system("OSI")
gc,miv,iv=stack_models(models)
ctx=O.GmmSystemCtx(O.default_cfg(),"OSI",gc,miv,iv,nthreads=8)
audio=synthetic_audio(4,16000)
pg=nes_params("OSI","targeted",samples_per_draw=10,seed=99,stream=3,target=1,threshold=0.05)
po=O.nes_params("OSI","targeted",ctx.S,samples_per_draw=10,target=1,threshold=0.05)
flg,gg,alg,scg=e.get_grad(pg,audio,it=5)
flo,go,alo,sco=O.get_grad(po,ctx.fn,ctx.ctx,audio,seed=99,it=5,stream=3)
print("loss",flg,flo,alg,alo)
d=np.abs(gg-go)
bad=np.where(~(d<1.0))[0]
print("nbad",bad.size, bad[:40], bad[-10:], (bad%4)[:40])
print("gpu",gg[bad[:8]],"ora",go[bad[:8]])
z=e.debug_noise(99,5,3,16000,5); zo=O.noise(99,5,3,16000,5)
print("noise equal", np.array_equal(z.view(np.uint32),zo.view(np.uint32)), np.abs(zo).max(), np.abs(z).max())
i=bad[0] if bad.size else 0
print("z at bad", z[:,i], zo[:,i])
