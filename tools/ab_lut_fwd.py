"""A/B of the first layer's forward on one box: bench.py's timed region with the matrix-pipe byte-store forward
(default) and with the dense NT product on the fp32 input (lut_fwd_min beyond every batch), alternating.
    python tools/ab_lut_fwd.py [rounds]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = ("import sys, runpy; sys.path.insert(0, %r); import dca_amd.config as c; o = c.EngineConfig.from_env.__func__\n"
        "def f(cls):\n    x = o(cls); x.lut_fwd_min = %%d; return x\n"
        "c.EngineConfig.from_env = classmethod(f)\n"
        "sys.argv = ['bench.py', '--steps', '96', '--warmup', '16', '--no-cpu-baseline']\n"
        "runpy.run_path(%r, run_name='__main__')\n" % (ROOT, os.path.join(ROOT, 'bench.py')))
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for name, v in (('lut', 1024), ('dense', 1 << 30)):
        out = subprocess.run([sys.executable, '-c', CODE % v], capture_output=True, text=True, cwd=ROOT).stdout
        line = [l for l in out.splitlines() if l.startswith('{')]
        if not line:
            print(name, 'no result', out[-400:]); continue
        j = json.loads(line[0])
        k = {x['kernel']: x['mean_ms'] for x in j.get('kernels', [])}
        print('%-5s ms_per_step %.4f  cells/s %.0f  heads %.4f  enc0_fwd(eager) %.4f  enc0_dW %.4f' % (
            name, j['ms_per_step'], j['value'], k.get('heads_fused', 0), k.get('gemm_enc0_fwd', 0), k.get('gemm_enc0_dW', 0)), flush=True)
