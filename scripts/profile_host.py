import cProfile, pstats, sys, io
sys.argv=['bench.py','--workload','allencahn','--steps','10','--warmup','2','--no-cpu-baseline']
sys.path.insert(0,'.')
import runpy
pr=cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s=io.StringIO()
ps=pstats.Stats(pr,stream=s).sort_stats('tottime')
ps.print_stats(28)
print(s.getvalue()[:6000])
