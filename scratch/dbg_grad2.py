import sys, numpy as np
sys.path.insert(0, '.')
from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import stack_models, synthetic_audio, synthetic_gmm_system
from oracle import oracle as O
ubm, spk = synthetic_gmm_system(n_speakers=3, C=256, D=72)
models=[ubm]+spk
e=Engine(0); e.load_gmm(models); e.set_system("OSI")
audio=synthetic_audio(4,16000)
for spd in [6,8,10,16,18,50]:
    half=spd//2
    pg=nes_params("OSI","targeted",samples_per_draw=spd,seed=99,stream=3,target=1,threshold=0.05)
    z=O.noise(99,5,3,audio.size,half).astype(np.float64).T.copy()
    a=e.get_grad(pg,audio,it=5)
    b=e.get_grad(pg,audio,it=5,noise_pos=z)
    bad=np.where(a[1]!=b[1])[0]
    print(spd,"philox-vs-explicit mismatches",bad.size,(bad%4)[:10], "max|g| philox",np.abs(a[1]).max(),"explicit",np.abs(b[1]).max())
