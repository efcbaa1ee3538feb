import os, sys, cProfile, pstats
sys.argv = ["compat_host_time.py"]
_R = "/root/repo"
sys.path.insert(0, os.path.join(_R, "tools"))
import runpy
ns = runpy.run_path(os.path.join(_R, "tools", "compat_host_time.py"))
m, data = ns["m"], ns["data"]
pr = cProfile.Profile(); pr.enable()
for _ in range(5): m.run(data)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
