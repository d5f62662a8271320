"""Static issue model of a gfx950 kernel's basic blocks (no GPU needed): how well are the matrix instructions spread among
the vector instructions of one wave?

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -Iinclude -o /tmp/k.s dca_amd/csrc/dcahip_heads.hip
    python tools/isa_issue_model.py /tmp/k.s heads_fused_p4_kernelILb1ELb0ELb1E [--min 40]

Per basic block (label to label / branch): instruction counts by kind and two cycle estimates for ONE wave on its SIMD,
from tools/microbench/mfma_valu_interleave.hip as measured on the MI355X (profiles/r05a_mfma_valu_interleave.txt):
  * a plain vector instruction issues in ~4.3 cycles, a transcendental in ~8.5, LDS / memory / scalar instructions ~4 / 4 / 1;
  * a v_mfma_f32_32x32x16_bf16 occupies the matrix pipe for 32 cycles and costs the issuing wave ~16 cycles of vector issue
    (its own slot + the slots it blocks); the next MFMA cannot start before the pipe is free.
`model` walks the block in order with these rules (an in-order wave, waits for data ignored: a lower bound that shows pipe
stalls from clustered MFMAs); `additive` = what the same instructions cost when the two kinds do not overlap at all.
"""
import re
import sys

VALU_C, TRANS_C, DS_C, VMEM_C, SALU_C, MFMA_ISSUE, MFMA_PIPE = 4.3, 8.5, 4.0, 4.0, 1.0, 16.0, 32.0
TRANS = ('v_exp_f32', 'v_log_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_sin_f32', 'v_cos_f32', 'v_rcp_iflag_f32')


def kind(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'mfma'
    if op.startswith(TRANS):
        return 'trans'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'ds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def blocks(lines):
    cur, name = [], 'entry'
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((';', '.')):
            if re.match(r'^\.LBB\d+_\d+:', t):
                if cur:
                    yield name, cur
                cur, name = [], t.split(':')[0]
            continue
        if re.match(r'^[A-Za-z_.$][\w.$]*:', t):
            continue
        op = t.split()[0]
        cur.append((op, t))
        if op.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc')):
            yield name, cur
            cur, name = [], name + "'"
    if cur:
        yield name, cur


def model(ins):
    t = pipe_free = 0.0
    stall = 0.0
    for op, _ in ins:
        k = kind(op)
        if k == 'mfma':
            if pipe_free > t:
                stall += pipe_free - t
                t = pipe_free
            pipe_free = t + MFMA_PIPE
            t += MFMA_ISSUE
        else:
            t += {'valu': VALU_C, 'trans': TRANS_C, 'ds': DS_C, 'vmem': VMEM_C, 'salu': SALU_C, 'wait': 0.0, 'other': 1.0}[k]
    return max(t, pipe_free), stall


def main():
    path, sym = sys.argv[1], sys.argv[2]
    mn = int(sys.argv[sys.argv.index('--min') + 1]) if '--min' in sys.argv else 40
    lines = open(path).read().split('\n')
    start = next(i for i, ln in enumerate(lines) if sym in ln and ln.rstrip().endswith(':') or (sym in ln and ':' in ln and '@' in ln))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    tot = {}
    print('%-12s %6s %6s %6s %6s %6s %6s %6s | %8s %8s %8s  %s' % ('block', 'valu', 'trans', 'mfma', 'ds', 'vmem', 'salu', 'lanes', 'model', 'additive', 'stall', 'scratch / readlane'))
    for name, ins in blocks(lines[start:end + 1]):
        c = {}
        for op, _ in ins:
            c[kind(op)] = c.get(kind(op), 0) + 1
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
        if len(ins) < mn:
            continue
        lanes = sum(1 for op, _ in ins if op.startswith(('v_readlane', 'v_writelane')))
        scr = sum(1 for op, _ in ins if op.startswith('scratch_'))
        est, stall = model(ins)
        add = (c.get('valu', 0) * VALU_C + c.get('trans', 0) * TRANS_C + c.get('ds', 0) * DS_C + c.get('vmem', 0) * VMEM_C +
               c.get('salu', 0) * SALU_C + c.get('mfma', 0) * MFMA_PIPE)
        print('%-12s %6d %6d %6d %6d %6d %6d %6d | %8.0f %8.0f %8.0f  %d / %d' % (name, c.get('valu', 0), c.get('trans', 0), c.get('mfma', 0),
              c.get('ds', 0), c.get('vmem', 0), c.get('salu', 0), lanes, est, add, stall, scr, lanes))
    print('static totals:', tot)


if __name__ == '__main__':
    main()
