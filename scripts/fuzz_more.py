"""More seeds for the two random-walk tests (deferred engine / level against eager ones): run on the GPU box,
   python scripts/fuzz_more.py   -> prints every failing (problem, seed) and the number of failures."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_fullsize as T
import tests.test_gpu_plugin as P
bad = 0
for n_, seeds in (("128", range(300, 310)), ("256", range(330, 335)), ("64", range(100, 140))):
  os.environ['PYSDC_FUZZ_N'] = n_
  for seed in seeds:
      for prob in ('heat_unforced', 'advdiff', 'heat_forced'):
          try:
              T.test_deferred_state_machine_random_walk.__wrapped__(prob, seed) if hasattr(T.test_deferred_state_machine_random_walk, '__wrapped__') else T.test_deferred_state_machine_random_walk(prob, seed)
          except Exception as e:
              bad += 1
              print('ENGINE FAIL', n_, prob, seed, str(e)[:400])
os.environ['PYSDC_FUZZ_N'] = '64'
for m_, seeds in (('2', range(400, 420)), ('5', range(420, 440)), ('6', range(440, 460)), ('8', range(460, 470))):
    os.environ['PYSDC_FUZZ_M'] = m_
    for seed in seeds:
        for prob in ('heat_unforced', 'advdiff', 'heat_forced'):
            try:
                T.test_deferred_state_machine_random_walk(prob, seed)
            except Exception as e:
                bad += 1
                print('ENGINE FAIL M', m_, prob, seed, str(e)[:400])
os.environ.pop('PYSDC_FUZZ_M')
for seed in range(100, 200):
    try:
        P.test_plugin_random_walk_deferred_vs_eager(seed)
    except Exception as e:
        bad += 1
        print('PLUGIN FAIL', seed, str(e)[:400])
for v in ('1', '2', '3'):   # the limit on virtual sweeps low enough for the walks to cross it
    os.environ['PYSDC_FUZZ_VIRTUAL'] = v
    for seed in range(500, 560):
        for prob in ('heat_unforced',):
            try:
                T.test_deferred_state_machine_random_walk(prob, seed)
            except Exception as e:
                bad += 1
                print('ENGINE FAIL VIRTUAL', v, prob, seed, str(e)[:400])
os.environ.pop('PYSDC_FUZZ_VIRTUAL')
print('failures', bad)
