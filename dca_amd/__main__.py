"""Command line of the reference (``dca input outputdir [flags]``, dca/__main__.py:18-154) on the
MI355X path: same positionals, flag names, defaults and output files.  The option table below
restates the reference's flag set; ``--hyper*`` runs dca_amd/hyper.py (the search space and outputs of
dca/hyper.py with hyperopt's TPE restated in dca_amd/tpe.py); ``--tensorboard`` is accepted for
command-line compatibility and ignored.
"""
import argparse
import sys

# (flags, kwargs) -- names / defaults as dca/__main__.py:31-136
_OPTIONS = [
    (('--normtype',), dict(type=str, default='zheng', help='size factor estimation: deseq | zheng (parsed, unused -- as in the reference)')),
    (('-t', '--transpose'), dict(dest='transpose', action='store_true', help='input is cell x gene (default: gene x cell)')),
    (('--testsplit',), dict(dest='testsplit', action='store_true', help='hold one fold out as a test set')),
    (('--type',), dict(type=str, default='nb-conddisp', help='autoencoder type (see dca_amd.network.AE_types)')),
    (('--threads',), dict(type=int, default=None, help='host threads of the native host stages (the reference sized TensorFlow CPU pools with it; the step runs on the GPU)')),
    (('-b', '--batchsize'), dict(type=int, default=32, help='batch size (default: 32)')),
    (('--sizefactors',), dict(dest='sizefactors', action='store_true', help='normalise means by library size (default)')),
    (('--nosizefactors',), dict(dest='sizefactors', action='store_false')),
    (('--norminput',), dict(dest='norminput', action='store_true', help='zero-mean normalise the input (default)')),
    (('--nonorminput',), dict(dest='norminput', action='store_false')),
    (('--loginput',), dict(dest='loginput', action='store_true', help='log-transform the input (default)')),
    (('--nologinput',), dict(dest='loginput', action='store_false')),
    (('-d', '--dropoutrate'), dict(type=str, default='0.0', help='dropout rate(s), comma separated')),
    (('--batchnorm',), dict(dest='batchnorm', action='store_true', help='batch normalisation (default)')),
    (('--nobatchnorm',), dict(dest='batchnorm', action='store_false')),
    (('--l2',), dict(type=float, default=0.0)),
    (('--l1',), dict(type=float, default=0.0)),
    (('--l2enc',), dict(type=float, default=0.0)),
    (('--l1enc',), dict(type=float, default=0.0)),
    (('--ridge',), dict(type=float, default=0.0, help='L2 penalty on the dropout probabilities')),
    (('--gradclip',), dict(type=float, default=5.0, help='clip gradient values (default: 5.0)')),
    (('--activation',), dict(type=str, default='relu')),
    (('--optimizer',), dict(type=str, default='RMSprop')),
    (('--init',), dict(type=str, default='glorot_uniform')),
    (('-e', '--epochs'), dict(type=int, default=300)),
    (('--earlystop',), dict(type=int, default=15)),
    (('--reducelr',), dict(type=int, default=10)),
    (('-s', '--hiddensize'), dict(type=str, default='64,32,64')),
    (('--inputdropout',), dict(type=float, default=0.0)),
    (('-r', '--learningrate'), dict(type=float, default=None)),
    (('--saveweights',), dict(dest='saveweights', action='store_true')),
    (('--no-saveweights',), dict(dest='saveweights', action='store_false')),
    (('--hyper',), dict(dest='hyper', action='store_true')),
    (('--hypern',), dict(dest='hypern', type=int, default=1000)),
    (('--hyperepoch',), dict(dest='hyperepoch', type=int, default=100)),
    (('--debug',), dict(dest='debug', action='store_true')),
    (('--tensorboard',), dict(dest='tensorboard', action='store_true')),
    (('--checkcounts',), dict(dest='checkcounts', action='store_true')),
    (('--nocheckcounts',), dict(dest='checkcounts', action='store_false')),
    (('--denoisesubset',), dict(dest='denoisesubset', type=str, help='file with gene names (one per line) to denoise')),
]

_DEFAULTS = dict(transpose=False, testsplit=False, saveweights=False, sizefactors=True, batchnorm=True,
                 checkcounts=True, norminput=True, hyper=False, debug=False, tensorboard=False,
                 loginput=True)


def build_parser():
    parser = argparse.ArgumentParser(prog='dca', description='Autoencoder')
    parser.add_argument('input', type=str, help='raw counts: TSV/CSV (gene x cell unless -t) or H5AD')
    parser.add_argument('outputdir', type=str, help='output directory')
    for flags, kw in _OPTIONS:
        parser.add_argument(*flags, **kw)
    parser.set_defaults(**_DEFAULTS)
    return parser


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    from .train import train_with_args
    train_with_args(args)


if __name__ == '__main__':
    main(sys.argv[1:])
