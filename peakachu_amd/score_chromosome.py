"""`peakachu score_chromosome` for the MI355X path
(peakachu/score_chromosome.py:3-71): same flags, same output."""
import os

import numpy as np

from . import io
from .forest import load_model
from .score_genome import build_chromosome, join_warm, warm_imports


def main(args):
    warm = warm_imports(getattr(args, "device", 0))
    try:
        return _main(args)
    finally:
        join_warm(warm)   # (an early error must not tear the interpreter down under a thread inside hipInit)


def _main(args):
    np.seterr(divide='ignore', invalid='ignore')
    if os.path.exists(args.output):
        os.remove(args.output)
    model = load_model(args.model)
    correct = False if args.clr_weight_name.lower() == 'raw' else args.clr_weight_name
    width = int((np.sqrt(model.feature_importances_.size) - 1) / 2)
    Lib = io.open_map(args.path)
    key = args.chrom
    label = 'chr' + key.lstrip('chr')  # the output label is always "chr"-prefixed (score_chromosome.py:37)
    X = build_chromosome(Lib, key, label, model, correct, args, width, getattr(args, "device", 0))
    result, R = X.score(thre=args.minimum_prob)
    X.writeBed(args.output, result, R)
