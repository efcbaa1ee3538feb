"""GPU: run the pre-training stage (lib/diode_pretraining.py) and report losses + speed.
usage: python tools/pretrain_run.py <n_layers> <layer_size> <epochs> [out.json]
The reference's documented results after 2000 epochs (diode_pretraining.py:196-200), 1N4148 (1U-1D):
2x4 1.34e-3/1.23e-3, 2x8 5.51e-5/2.49e-4, 2x16 7.98e-6/9.49e-5, 4x4 6.38e-4/8.48e-4, 4x8 4.43e-5/2.24e-4."""
import json, os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "differentiable-wdfs_amd", "lib"))
import torch
import diode_pretraining as dp
from diode_config import diode_1n4148_1u1d
from model_utils import save_model

n_layers, size, epochs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t0 = time.time()
marks = []


def log(epoch, loss):
    if epoch % max(1, epochs // 20) == 0 or epoch == epochs - 1:
        marks.append((epoch, loss, time.time() - t0))
        print(f"epoch {epoch}: mean batch loss {loss:.4e}  ({time.time() - t0:.1f} s)", flush=True)


model, stats = dp.pretrain(diode_1n4148_1u1d, n_layers, size, epochs=epochs, log=log)
torch.cuda.synchronize()
dt = time.time() - t0
steps = epochs * 625
res = {"net": f"{n_layers}x{size}", "epochs": epochs, "seconds": dt, "steps_per_s": steps / dt,
       "mse_esr_before": stats["before"], "mse_esr_after": stats["after"], "marks": marks}
print(json.dumps(res))
if len(sys.argv) > 4:
    json.dump(res, open(sys.argv[4], "w"))
    save_model(model, sys.argv[4].replace(".json", "_model.json"))
