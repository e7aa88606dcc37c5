"""Golden statistics of the reference's training-mask generator (run here, with /root/reference importable; writes
tests/golden/mask_stats.json).  ``misc.get_masks_for_training`` (/root/reference/misc.py:13-68) is called 45 000 times
with ``skimage.draw.random_shapes`` stubbed (scikit-image is not installed: the stub returns an all-background image, so the
DECISIONS - which stage is open, whether a spatial mask is used - are the unmodified reference's, only the shapes are not
drawn).  Recorded: the joint counts of (open stage counted from the deep end, spatial) and the list-index layout check."""
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_stubs  # noqa: E402


def main():
    models, lossfunction, model_wrapper, misc = reference_stubs.import_reference()
    import skimage.draw as skd                                           # the stub module installed by reference_stubs
    calls = {"n": 0}

    def random_shapes(shape, **kw):
        calls["n"] += 1
        return np.full(tuple(shape) + (3,), 255, dtype=np.uint8), None
    skd.random_shapes = random_shapes
    misc.random_shapes = random_shapes
    random.seed(20260)
    np.random.seed(20260)
    n = 45000
    counts = {}
    for _ in range(n):
        before = calls["n"]
        masks = misc.get_masks_for_training()
        spatial = calls["n"] > before
        # list index i <-> VGG feature i; the open stage counted from the deep end is 6 - (last all-ones level)
        ones = [i for i, m in enumerate(masks) if bool((m == 1).all())]
        stage = 6 - max(ones) if not spatial else 6 - max(i for i in ones if all(float(masks[j].max()) == 0 for j in range(i + 1, 7)))
        counts[(stage, spatial)] = counts.get((stage, spatial), 0) + 1
    out = {"n": n, "counts": [[s, int(sp), c] for (s, sp), c in sorted(counts.items())],
           "source": "/root/reference/misc.py:13-68 get_masks_for_training(), random.seed(20260), np.random.seed(20260)"}
    json.dump(out, open(os.path.join(HERE, "mask_stats.json"), "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
