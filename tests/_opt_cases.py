"""Shared body of the optimizer / regulariser fit-parity tests (CPU with the oracle-backed ops, GPU
with HipOps): the engine's fit loop against oracle.net_np.fit, same weights, same batch order."""
import numpy as np

from helpers import make_problem, make_engine
from oracle import net_np as N

OPTIMIZERS = ['SGD', 'Adagrad', 'Adadelta', 'Adam', 'Adamax', 'Nadam']
REG_CASES = [(1e-4, 2e-4, 0., 0.), (1e-4, 0., 3e-4, 2e-4)]


def run_fit_parity(ops, optimizer='RMSprop', reg=(0., 0., 0., 0.), ae_type='zinb-conddisp', rtol=5e-4):
    from dca_amd.train import fit_engine, KERAS_DEFAULT_LR
    n, G, hs = 75, 20, (6, 3, 6)
    X, Y, sf, p = make_problem(n, G, hs, ae_type, True, seed=4)
    ref = N.OracleAE(ae_type, {k: np.asarray(v, np.float64).copy() for k, v in p.items()}, hs, True, 0.0, reg)
    rh = N.fit(ref, X.astype(np.float64), Y.astype(np.float64), sf.astype(np.float64), epochs=3,
               batch_size=16, shuffle_rng=np.random.RandomState(9), reduce_lr=1, early_stop=0,
               optimizer=optimizer)
    eng = make_engine(ops, ae_type, G, hs, True, 0.0, p, X, Y, sf)
    eng.set_optimizer(optimizer)
    eng.set_regularizers(*reg)
    n_train = int(n * 0.9)
    h = fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=3, batch_size=16,
                   shuffle_rng=np.random.RandomState(9), reduce_lr=1, early_stop=0,
                   learning_rate=KERAS_DEFAULT_LR[optimizer.lower()])
    np.testing.assert_allclose(h.history['loss'], rh['loss'], rtol=rtol)
    np.testing.assert_allclose(h.history['val_loss'], rh['val_loss'], rtol=rtol)
    np.testing.assert_allclose(h.history['lr'], rh['lr'], rtol=1e-7)
    newp = eng.get_params()
    for k in ref.p:
        if k[0] == 'b' and k[1:].isdigit():
            continue      # bias in front of BatchNorm: its gradient is round-off noise, which the
                          # sign-normalising optimizers (Adam, Adamax, Adadelta) turn into O(lr) steps
        np.testing.assert_allclose(newp[k], ref.p[k], rtol=5e-3, atol=3e-5, err_msg=k)
