"""Generates tests/golden/fit_c2_oracle.npz: the oracle's Keras-fit trajectories on BASELINE configs[1]
(2 000 x 1 000, 64-32-64, batch 32, 2 epochs) for a SET of problem seeds, in fp64 (truth) and in fp32
(an fp32 evaluation of the same restatement -- what the reference's fp32 TensorFlow graph is to its
own exact arithmetic).  tests/test_engine_gpu.py::test_fit_epoch_losses_match_oracle compares the
MI355X fit loop with these numbers; tests/test_oracle_golden.py re-derives a sample of them on the CPU
so that the fixture cannot go stale silently.

    python tests/golden/make_fit_c2_golden.py          (about 3 minutes on 8 cores)

Why a set of seeds: multi-epoch fp32 trajectories are chaotic at ReLU boundaries (a hidden
pre-activation within ~1e-7 of zero flips its mask under any re-association).  The fp32 oracle itself
leaves the fp64 trajectory at one seed in ten (seed 7 of zinb-conddisp: val_loss 3e-4 apart, all others
~1e-8), so parity of a fit loop is a statement about a distribution, not about one seed.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

N_CELLS, N_GENES, HIDDEN, EPOCHS, BATCH, SHUFFLE_SEED = 2000, 1000, (64, 32, 64), 2, 32, 5
SEEDS = {'zinb-conddisp': list(range(10)), 'nb': list(range(10)), 'zinb': [0, 1, 2, 3], 'nb-conddisp': [0, 1, 2, 3],
         'poisson': [0, 1, 2, 3], 'normal': [0, 1, 2, 3]}
N_PREDICT = 16          # cells whose outputs after training are stored (first seed of every type)


def oracle_fit(ae_type, seed, dtype):
    from helpers import make_problem, oracle_net
    from oracle import net_np as N
    X, Y, sf, p = make_problem(N_CELLS, N_GENES, HIDDEN, ae_type, True, seed=seed)
    net = oracle_net(ae_type, p, HIDDEN, True, dtype=dtype)
    h = N.fit(net, X.astype(dtype), Y.astype(dtype), sf.astype(dtype), epochs=EPOCHS, batch_size=BATCH,
              shuffle_rng=np.random.RandomState(SHUFFLE_SEED))
    out = net.predict(X[:N_PREDICT].astype(dtype), sf[:N_PREDICT].astype(dtype))
    return h, out


def main():
    res = {}
    for ae_type, seeds in SEEDS.items():
        for seed in seeds:
            for dtype, tag in ((np.float64, 'f64'), (np.float32, 'f32')):
                h, out = oracle_fit(ae_type, seed, dtype)
                key = '%s/%d/%s' % (ae_type, seed, tag)
                res[key + '/loss'] = np.asarray(h['loss'], np.float64)
                res[key + '/val_loss'] = np.asarray(h['val_loss'], np.float64)
                if seed == seeds[0] and tag == 'f64':
                    for k, v in out.items():
                        if v is not None:
                            res[key + '/out_' + k] = np.asarray(v, np.float32)
            d = [abs(a / b - 1) for q in ('loss', 'val_loss')
                 for a, b in zip(res['%s/%d/f32/%s' % (ae_type, seed, q)], res['%s/%d/f64/%s' % (ae_type, seed, q)])]
            print(ae_type, seed, 'fp32 oracle vs fp64 oracle, rel:', ['%.1e' % x for x in d], flush=True)
    np.savez_compressed(os.path.join(HERE, 'fit_c2_oracle.npz'), **res)


if __name__ == '__main__':
    main()
