"""python tools/run_with_watchdog.py <seconds> <script> [args...]: runs the script; if it is still
running after <seconds>, dumps every thread's Python stack to stderr and exits (so that a hang on
the GPU box costs seconds of budget, not the whole call)."""
import faulthandler
import runpy
import sys

secs = int(sys.argv[1])
faulthandler.enable()          # a GPU memory fault aborts the process: show where Python was
faulthandler.dump_traceback_later(secs, exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
