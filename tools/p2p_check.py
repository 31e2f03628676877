"""Run tests/test_gpu_p2p.py's worker in two plain subprocesses and show their output (diagnostic)."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, queue
sys.path.insert(0, %r)
from tests.test_gpu_p2p import _worker
class Q:
    def put(self, x): print("RESULT", x[0], x[2], x[4], float(x[1].abs().sum()))
_worker(int(sys.argv[1]), 2, int(sys.argv[2]), Q())
''' % REPO
procs = [subprocess.Popen([sys.executable, "-c", code, str(r), "29611"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
for r, p in enumerate(procs):
    out, _ = p.communicate(timeout=200)
    print("---- rank", r, "rc", p.returncode)
    print(out[-3000:])
