"""Regenerates tests/golden/biochemists.npz from the reference's R-generated fixtures.

Run in the build container only (needs /root/reference):
    python tests/golden/make_biochemists_golden.py

Sources (reference repo, data/): biochemists.tsv (pscl bioChemists design matrix, 915 x 6),
biochemists-{nb,zinb}-coef.tsv (MASS::glm.nb / pscl::zeroinfl MLEs, data/biochemists.R:16-42),
biochemists-{nb,zinb}-predictions.tsv (R's predict()).  The fixtures travel as one small
binary so the GPU box (which has no /root/reference) can run the known-answer tests.
"""
import os
import numpy as np
import pandas as pd

REF = '/root/reference/data'
HERE = os.path.dirname(os.path.abspath(__file__))

tab = pd.read_csv(os.path.join(REF, 'biochemists.tsv'), sep='\t')
zc = pd.read_csv(os.path.join(REF, 'biochemists-zinb-coef.tsv'), sep='\t')
nc = pd.read_csv(os.path.join(REF, 'biochemists-nb-coef.tsv'), sep='\t')
zp = pd.read_csv(os.path.join(REF, 'biochemists-zinb-predictions.tsv'), sep='\t')
npred = pd.read_csv(os.path.join(REF, 'biochemists-nb-predictions.tsv'), sep='\t')

assert list(tab.columns) == ['art', 'fem', 'mar', 'kid5', 'phd', 'ment']
assert list(zc['coef']) == ['intercept', 'fem', 'mar', 'kid5', 'phd', 'ment', 'theta']
assert list(nc['coef']) == ['intercept', 'fem', 'mar', 'kid5', 'phd', 'ment', 'theta']

np.savez_compressed(
    os.path.join(HERE, 'biochemists.npz'),
    table=tab.values.astype(np.float64),            # [915, 6]: art + 5 covariates
    columns=np.array(tab.columns.tolist()),
    zinb_count_coef=zc['count'].values[:6].astype(np.float64),   # beta (intercept first)
    zinb_zero_coef=zc['zero'].values[:6].astype(np.float64),     # gamma
    zinb_theta=np.float64(zc['count'].values[6]),
    nb_coef=nc['val'].values[:6].astype(np.float64),
    nb_theta=np.float64(nc['val'].values[6]),
    zinb_pred_zero=zp['zero'].values.astype(np.float64),
    zinb_pred_count=zp['count'].values.astype(np.float64),
    nb_pred_count=npred['count'].values.astype(np.float64),
)
print('wrote', os.path.join(HERE, 'biochemists.npz'))
