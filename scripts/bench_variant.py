"""Run bench.py against another build of the library (A/B measurements of compile-time knobs).
    python scripts/bench_variant.py raytracer-go_amd/csrc/variants/librtgo_f16.so [bench args]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
backend = importlib.import_module("raytracer-go_amd.backend")
backend.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
