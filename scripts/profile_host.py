"""cProfile of a bench.py run (host side): python scripts/profile_host.py [bench arguments]"""
import cProfile, pstats, sys, io
args = sys.argv[1:] or ['--workload', 'allencahn', '--steps', '10', '--warmup', '2']
sys.argv = ['bench.py'] + args + ['--no-cpu-baseline']
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
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
