"""bench.py's timed region with the product library and with experiment builds of it, alternating on one box.
    [BENCH_ARGS="--workload c5"] python tools/ab_heads_lib.py dca_amd/csrc/libdcahip_x.so [more.so ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = ("import sys, runpy; sys.path.insert(0, %r)\n"
        "lib = %%r\n"
        "if lib:\n    from dca_amd import build as b; b.LIB = lib; b.needs_build = lambda: False\n"
        "import os\nsys.argv = ['bench.py', '--no-cpu-baseline'] + (os.environ['BENCH_ARGS'].split() if os.environ.get('BENCH_ARGS') else ['--steps', '96', '--warmup', '16'])\n"
        "runpy.run_path(%r, run_name='__main__')\n" % (ROOT, os.path.join(ROOT, 'bench.py')))
libs = [''] + [os.path.abspath(a) for a in sys.argv[1:]]
for r in range(2):
    for lib in libs:
        out = subprocess.run([sys.executable, '-c', CODE % lib], capture_output=True, text=True, cwd=ROOT, stdin=subprocess.DEVNULL).stdout
        line = [l for l in out.splitlines() if l.startswith('{')]
        if not line:
            print(os.path.basename(lib) or 'product', 'no result'); continue
        j = json.loads(line[0])
        k = {x['kernel']: x['mean_ms'] for x in j.get('kernels', [])}
        print('%-22s ms_per_step %.4f  cells/s %.0f  %s' % (os.path.basename(lib) or 'product', j['ms_per_step'], j['value'], ' '.join('%s %.4f' % (x['kernel'], x['mean_ms']) for x in j.get('kernels', [])[:6])), flush=True)
