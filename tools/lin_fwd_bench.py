import sys, time, torch
sys.path.insert(0, "/root/repo/differentiable-wdfs_amd/lib")
import tf_wdf as wdf
from wdf_hip import binding
FS=48000
B,T=8192,4096
x=torch.randn((B,T),device="cuda")
Vs=wdf.IdealVoltageSource(); R1,C1=wdf.Resistor(1000,True),wdf.Capacitor(1e-6,FS,True)
circ=wdf.Circuit(wdf.Inverter(wdf.Series(R1,C1)),Vs,C1)
with torch.no_grad():
    for _ in range(3): y=circ(x)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): y=circ(x)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
    e0,e1=binding.Event(),binding.Event(); binding.Event.bracket_next(e0,e1); y=circ(x); torch.cuda.synchronize()
print(f"RC lowpass forward through circ(x): {dt*1e3:.3f} ms per call ({B*T/dt/1e9:.1f} G samples/s); chunk kernel {e0.elapsed_ms(e1):.4f} ms")
