import faulthandler, sys, os, runpy
faulthandler.dump_traceback_later(int(os.environ.get("DUMP_AFTER", "60")), exit=True)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
