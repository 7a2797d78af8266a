"""run a tools/ script against another library build: python tools/run_with_lib.py <lib.so> <script.py> [args...]"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rls_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1]); L._lib = None
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
