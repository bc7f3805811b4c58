"""database_<iter>.npz (bunmpc_amd/dataset.py, written where h5py is absent) -> database_<iter>.hdf5 in the layout of the
reference's save_dataset (data sets states / vc_goals / cc_goals / actions).  Needs h5py.  usage: npz_to_hdf5.py file.npz"""
import sys

import numpy as np


def main(path):
    import h5py
    out = path[:-4] + ".hdf5"
    with np.load(path) as z, h5py.File(out, "w") as hf:
        for k in ("states", "vc_goals", "cc_goals", "actions"):
            hf.create_dataset(k, data=z[k])
    print(out)


if __name__ == "__main__":
    main(sys.argv[1])
