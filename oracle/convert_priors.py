"""One-off conversion (container-only tooling) of the reference's MATLAB v7.3 prior files to .npz
fixtures under tests/golden/, using the repo's own minimal HDF5 reader (no h5py here).
They are data files of the reference (learned fixation statistics), kept as fixtures so the
reader and `priors.get_bias` can be tested where /root/reference does not exist."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iip_uavsal_saliency_amd import matio  # noqa: E402

if __name__ == "__main__":
    for name in ("gauss_priors", "UAV2_ob_priors_train", "AVS1K_ob_priors_train"):
        a = matio.loadmat("/root/reference/%s.mat" % name)["PriorMaps"].astype(np.float32)
        out = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(out, PriorMaps=a)
        print(name, a.shape, "%.0f KB" % (os.path.getsize(out) / 1024))
