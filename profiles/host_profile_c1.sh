#!/bin/bash
# Where the host's time goes in the smallest leg (TicTacToe, one game, 25 simulations per move): cProfile of the bench process
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--board', '3', '--playouts', '25', '--games', '1', '--lanes', '1', '--steps', '400', '--warmup', '50', '--regions', '1',
            '--no-configs', '--no-fill', '--no-games-leg', '--no-cpu-baseline', '--timeline', '0', '--eager-every', '0']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumtime').print_stats('rlzero_amd|bench.py', 45)
print(s.getvalue()[:9000])
"
