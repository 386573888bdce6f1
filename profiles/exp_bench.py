"""bench.py against an experiment build of the library (make -C project3-cuda-path-tracer_amd/csrc exp EXP=n):
    python profiles/exp_bench.py <n> [bench.py arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pt = ge.load_package()
pt.LIB_PATH = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "csrc", "libpt_amd_exp%s.so" % sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import runpy
runpy.run_path(sys.argv[0], run_name="__main__")
