"""development aid: run one soak_mid seed and dump the Python stack if it takes longer than 40 s"""
import faulthandler, sys, runpy
faulthandler.dump_traceback_later(40, exit=True)
sys.argv = ['soak_mid.py', sys.argv[1], str(int(sys.argv[1]) + 1)]
runpy.run_path(__file__.replace('where.py', 'soak_mid.py'), run_name='__main__')
