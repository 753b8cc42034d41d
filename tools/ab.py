"""within-probe A/B of library variants: python tools/ab.py v0 v1 ...  (interleaved rounds)"""
import os, subprocess, sys, json
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
print("RESULT", min(ts), sorted(ts)[len(ts)//2])
''' % root
for rnd in range(2):
    for v in sys.argv[1:]:
        env = dict(os.environ, AFX_LIB=os.path.join(root, "audioflux_amd", "lib", "variants", f"libafx_{v}.so"))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print(rnd, v, line[0] if line else out.stderr[-300:], flush=True)
