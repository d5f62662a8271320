"""CLI surface: flag names / defaults of dca/__main__.py and the end-to-end file outputs
(mean.tsv gene x cell, latent.tsv cell x dim, dispersion.tsv, dropout.tsv; network.py:213-231,
407-421)."""
import os

import numpy as np
import pandas as pd

from conftest import synth_counts
from dca_amd.__main__ import parse_args, main
from dca_amd.network import override_ops
from oracle.cpu_ops import CpuRefOps


def test_flag_defaults_match_reference():
    a = parse_args(['in.tsv', 'out'])
    assert (a.type, a.batchsize, a.epochs, a.earlystop, a.reducelr, a.hiddensize) == \
        ('nb-conddisp', 32, 300, 15, 10, '64,32,64')
    assert (a.gradclip, a.optimizer, a.activation, a.init, a.learningrate) == \
        (5.0, 'RMSprop', 'relu', 'glorot_uniform', None)
    assert a.sizefactors and a.norminput and a.loginput and a.batchnorm and a.checkcounts
    assert not (a.transpose or a.testsplit or a.saveweights or a.hyper or a.debug or a.tensorboard)
    assert (a.l1, a.l2, a.l1enc, a.l2enc, a.ridge, a.inputdropout, a.dropoutrate) == (0, 0, 0, 0, 0, 0, '0.0')
    b = parse_args(['in.tsv', 'out', '--type', 'zinb-conddisp', '-b', '64', '-s', '16,2,16', '--nobatchnorm',
                    '--nosizefactors', '--nonorminput', '--nologinput', '--nocheckcounts', '-t'])
    assert (b.type, b.batchsize, b.hiddensize) == ('zinb-conddisp', 64, '16,2,16')
    assert not (b.batchnorm or b.sizefactors or b.norminput or b.loginput or b.checkcounts) and b.transpose


def test_cli_end_to_end(tmp_path):
    n, G = 70, 24
    y = synth_counts(n, G, 5)
    genes = ['g%d' % i for i in range(G)]; cells = ['c%d' % i for i in range(n)]
    f = str(tmp_path / 'counts.tsv')
    pd.DataFrame(y.T.astype(int), index=genes, columns=cells).to_csv(f, sep='\t')
    out = str(tmp_path / 'res')
    with override_ops(CpuRefOps):
        main([f, out, '--type', 'zinb-conddisp', '-e', '2', '-s', '8,2,8', '--saveweights'])
    for name in ('mean.tsv', 'latent.tsv', 'dispersion.tsv', 'dropout.tsv', 'model.pickle', 'weights.npz'):
        assert os.path.exists(os.path.join(out, name)), name
    mean = pd.read_csv(os.path.join(out, 'mean.tsv'), sep='\t', index_col=0)
    assert mean.shape == (G, n) and list(mean.index) == genes and list(mean.columns) == cells
    lat = pd.read_csv(os.path.join(out, 'latent.tsv'), sep='\t', index_col=0, header=None)
    assert lat.shape == (n, 2)
    # dispersion / dropout are written gene x cell WITHOUT a header row (network.py:413-421 pass
    # no rownames, so after the transpose there are no column names)
    drop = pd.read_csv(os.path.join(out, 'dropout.tsv'), sep='\t', index_col=0, header=None)
    assert drop.shape == (G, n) and list(drop.index) == genes and ((drop.values >= 0) & (drop.values <= 1)).all()


def test_cli_hyper_search(tmp_path):
    """dca --hyper (dca/hyper.py): the reference's space, TPE proposals (random while fewer than 20 trials have finished:
    tests/test_tpe_cpu.py holds the algorithm), best.json + trials.pickle."""
    import json
    import pickle
    from dca_amd import hyper as H
    rng = np.random.RandomState(0)
    draws = [H.sample(rng) for _ in range(200)]
    assert all(1e-3 <= d['model']['lr'] <= 1e-2 and 1e-7 <= d['model']['ridge'] <= 1e-1 for d in draws)
    assert all(0 <= d['model']['dropout'] <= 0.7 and 0 <= d['model']['input_dropout'] <= 0.8 for d in draws)
    assert {d['model']['hidden_size'] for d in draws} == set(H.HIDDEN_SIZES)
    assert {d['model']['activation'] for d in draws} == set(H.ACTIVATIONS)
    assert {d['model']['aetype'] for d in draws} == {'zinb', 'zinb-conddisp'}
    n, G = 60, 16
    y = synth_counts(n, G, 7)
    f = str(tmp_path / 'counts.tsv')
    pd.DataFrame(y.T.astype(int), index=['g%d' % i for i in range(G)], columns=['c%d' % i for i in range(n)]).to_csv(f, sep='\t')
    out = str(tmp_path / 'res')
    with override_ops(CpuRefOps):
        main([f, out, '--hyper', '--hypern', '6', '--hyperepoch', '2'])
    res = os.path.join(out, 'hyperopt_results')
    best = json.load(open(os.path.join(res, 'best.json')))
    trials = pickle.load(open(os.path.join(res, 'trials.pickle'), 'rb'))
    assert len(trials) == 6 and {'loss', 'tid', 'data', 'model'} <= set(best)
    ok = [t for t in trials if t['status'] == 'ok']
    assert ok and best['loss'] == min(t['loss'] for t in ok)
    assert all(t['status'] == 'ok' for t in trials), [t.get('error') for t in trials]    # the whole space runs
    assert not os.path.exists(os.path.join(out, 'mean.tsv'))        # hyper() runs and exits (train.py:119-122)
