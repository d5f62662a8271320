"""Hyper-parameter search of dca/hyper.py:14-112 (`dca --hyper`) without kopt / hyperopt.

Same search space (hyper.py:19-43), same objective construction (data_fn: normalize with the sampled
flags, hyper.py:45-57; model_fn: AE_types[aetype] with the sampled architecture / regularisation,
RMSprop(lr, clipvalue=5), hyper.py:59-83; a random 20 % of the cells held out, the loss on them is minimised,
hyper.py:85-95), same outputs (`<outputdir>/hyperopt_results/best.json`, `trials.pickle`), same proposal
algorithm: hyperopt's TPE (fmin(algo=tpe.suggest), hyper.py:97-104) restated in dca_amd/tpe.py with that module's
defaults -- 20 random start-up trials, then proposals that maximise l(x) / g(x) of the adaptive Parzen estimators of the
good and the other trials; the random streams are numpy's (RandomState(42)), so individual proposals differ from a
hyperopt run.  A trial that fails (a
diverged loss, say) is recorded as failed and skipped, as fmin(catch_eval_exceptions=True) does
(hyper.py:99-104).  best.json holds the chosen VALUES (the
reference writes hyperopt's choice indices and carries a TODO about it, hyper.py:109).  As in the
reference the input is read with transpose=args.transpose (hyper.py:15-17: NOT the `not args.transpose`
of the training pipeline, train.py:124-127), i.e. `--hyper` expects cell x gene unless -t is given.
"""
import json
import os
import pickle

import numpy as np

from . import io
from . import tpe as _tpe
from .network import AE_types
from .train import train

HIDDEN_SIZES = ((64, 32, 64), (32, 16, 32), (64, 64), (32, 32), (16, 16), (16,), (32,), (64,), (128,))
ACTIVATIONS = ('relu', 'selu', 'elu', 'PReLU', 'linear', 'LeakyReLU')
AE_CHOICES = ('zinb', 'zinb-conddisp')
KOPT_PATIENCE = 10          # kopt.CompileFN's default early-stopping patience
VALID_SPLIT = 0.2           # hyper.py:90


# hyper.py:19-43, flat: hyperopt label -> distribution (choices by index, as hyperopt records them)
SPACE = {
    'd_norm_log': ('choice', 2), 'd_norm_zeromean': ('choice', 2), 'd_norm_sf': ('choice', 2),
    'm_lr': ('loguniform', 1e-3, 1e-2), 'm_ridge': ('loguniform', 1e-7, 1e-1), 'm_l1_enc_coef': ('loguniform', 1e-7, 1e-1),
    'm_hiddensize': ('choice', len(HIDDEN_SIZES)), 'm_activation': ('choice', len(ACTIVATIONS)),
    'm_aetype': ('choice', len(AE_CHOICES)), 'm_batchnorm': ('choice', 2),
    'm_do': ('uniform', 0.0, 0.7), 'm_input_do': ('uniform', 0.0, 0.8),
}
_BOOL = (True, False)                   # hp.choice(label, (True, False)): index 0 is True


def to_params(v):
    """A proposal over SPACE (hyperopt's labels, choice indices) -> the nested arguments of data_fn / model_fn."""
    return {
        'data': {'norm_input_log': _BOOL[v['d_norm_log']], 'norm_input_zeromean': _BOOL[v['d_norm_zeromean']],
                 'norm_input_sf': _BOOL[v['d_norm_sf']]},
        'model': {'lr': float(v['m_lr']), 'ridge': float(v['m_ridge']), 'l1_enc_coef': float(v['m_l1_enc_coef']),
                  'hidden_size': HIDDEN_SIZES[v['m_hiddensize']], 'activation': ACTIVATIONS[v['m_activation']],
                  'aetype': AE_CHOICES[v['m_aetype']], 'batchnorm': _BOOL[v['m_batchnorm']],
                  'dropout': float(v['m_do']), 'input_dropout': float(v['m_input_do'])},
    }


def sample(rng):
    """One uniform draw from the space of hyper.py:19-43 (what the start-up trials of the search use)."""
    t = _tpe.TPE(SPACE)
    t.rng = rng
    return to_params(t.random())


def evaluate(adata, params, epochs, debug=False, seed=0):
    """Trains one configuration; returns the best held-out loss and the history."""
    d, m = params['data'], params['model']
    # kopt.CompileFN(valid_split=.2) holds out a RANDOM fifth of the cells (hyper.py:85-95); train() takes the
    # validation rows from the tail (Keras validation_split), so the cells are permuted once -- the same permutation
    # for every trial (seed 42): trials are ranked on the same held-out cells
    order = np.random.RandomState(42).permutation(adata.n_obs)
    ad = io.normalize(adata[order].copy(), size_factors=d['norm_input_sf'], logtrans_input=d['norm_input_log'],
                      normalize_input=d['norm_input_zeromean'])
    net = AE_types[m['aetype']](input_size=ad.n_vars, hidden_size=m['hidden_size'], l2_coef=0.0, l1_coef=0.0,
                                l2_enc_coef=0.0, l1_enc_coef=m['l1_enc_coef'], ridge=m['ridge'],
                                hidden_dropout=m['dropout'], input_dropout=m['input_dropout'],
                                batchnorm=m['batchnorm'], activation=m['activation'], init='glorot_uniform',
                                debug=debug)
    net.seed = seed
    net.build()
    hist = train(ad, net, optimizer='RMSprop', learning_rate=m['lr'], epochs=epochs, reduce_lr=None,
                 early_stop=KOPT_PATIENCE, batch_size=32, clip_grad=5.0, validation_split=VALID_SPLIT, verbose=False)
    val = [v for v in hist.history.get('val_loss', []) if np.isfinite(v)]
    if not val:
        raise FloatingPointError('no finite validation loss')
    return float(np.min(val)), hist.history


def hyper(args):
    adata = io.read_dataset(args.input, transpose=args.transpose, test_split=False)
    output_dir = os.path.join(args.outputdir, 'hyperopt_results')
    os.makedirs(output_dir, exist_ok=True)
    search = _tpe.TPE(SPACE, seed=42)
    trials, best, history = [], None, []
    for t in range(int(args.hypern)):
        vals = search.suggest(history)
        params = to_params(vals)
        rec = {'tid': t, 'params': params, 'vals': vals, 'status': 'ok'}
        try:
            rec['loss'], rec['history'] = evaluate(adata, params, int(args.hyperepoch), getattr(args, 'debug', False), seed=t)
        except Exception as e:          # fmin(catch_eval_exceptions=True)
            rec['status'], rec['error'] = 'fail', '%s: %s' % (type(e).__name__, e)
        trials.append(rec)
        history.append((vals, rec.get('loss') if rec['status'] == 'ok' else None))
        if rec['status'] == 'ok' and (best is None or rec['loss'] < best['loss']):
            best = rec
        print('dca: hyper trial %d/%d: %s' % (t + 1, args.hypern, ('loss %.6f' % rec['loss']) if rec['status'] == 'ok'
                                              else rec['error']))
    with open(os.path.join(output_dir, 'trials.pickle'), 'wb') as f:
        pickle.dump(trials, f)
    best_out = {} if best is None else dict(loss=best['loss'], tid=best['tid'], **best['params'])
    with open(os.path.join(output_dir, 'best.json'), 'wt') as f:
        json.dump(best_out, f, sort_keys=True, indent=4)
    print(best_out)
    return best_out
