"""A/B of runtime switches (environment variables) of the shipped library on the headline
kernel: python tools/ab_env.py "" "AFX_PAIR=1" ...   (interleaved rounds, min / median ms)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
import audioflux_amd as af
x = 0.1*torch.randn((1000, 480000), device="cuda")
bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
             scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
bft.set_result_type(1)
out = torch.empty((1000, 934, 128), device="cuda")
for _ in range(3): bft.bft_device(x, out_real=out)
torch.cuda.synchronize()
ts=[]
for r in range(5):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): bft.bft_device(x, out_real=out)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/5)
print("RESULT", min(ts), sorted(ts)[len(ts)//2], float(out.double().sum()))
''' % root
for rnd in range(2):
    for v in sys.argv[1:]:
        env = dict(os.environ)
        for kv in v.split():
            k, val = kv.split("=")
            env[k] = val
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print(rnd, repr(v), line[0] if line else out.stderr[-400:], flush=True)
