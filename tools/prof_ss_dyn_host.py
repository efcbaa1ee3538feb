import cProfile, pstats, sys, io, runpy
sys.argv=['tools/ss_dyn_bench.py']
pr=cProfile.Profile(); pr.enable()
try:
    runpy.run_path('tools/ss_dyn_bench.py', run_name='__main__')
finally:
    pr.disable()
    s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('tottime').print_stats(45); print(s.getvalue()[:9000])
