"""The data set the reference's trainers read (`ISL/examples/iterative_algorithm/database.py:9-146`,
`data_collection.py:93-124`): a ring buffer of rows `states (43)`, `vc_goals (5)`, `cc_goals (3 n_eff goal_horizon)`,
`actions (12)` saved as HDF5 data sets of those names (`database_<iter>.hdf5`).

h5py is not part of this image, so `save()` writes the same four arrays to `database_<iter>.npz` unless h5py imports;
`tools/npz_to_hdf5.py` turns such a file into the reference's HDF5 on a machine that has h5py.  The reference also pickles
its hydra config next to the data, once (`config.pkl` = `pickle.dump(OmegaConf.to_container(cfg, resolve=True))`,
`data_collection.py:116-122`: a plain dict of builtins): written here the same way with the standard library's pickle, plus a
human-readable `config.json` beside it.  A hand-rolled HDF5 container stays unbuilt: nothing in this image can read one back,
so it could not be checked (EXPERIMENTS.md 10).
"""
import json
import os
import pickle

import numpy as np

STATE_WIDTH, VC_GOAL_WIDTH, ACTION_WIDTH = 43, 5, 12
GAIT_VALUE = {"trot": 1.0, "jump": 2.0, "bound": 3.0}        # utils.py:268-290 (anything else: 0)


class Database:
    def __init__(self, limit, n_eff=4, goal_horizon=1):
        self.limit, self.start, self.length = int(limit), 0, 0
        self.states = np.zeros((self.limit, STATE_WIDTH))
        self.vc_goals = np.zeros((self.limit, VC_GOAL_WIDTH))
        self.cc_goals = np.zeros((self.limit, 3 * n_eff * goal_horizon))
        self.actions = np.zeros((self.limit, ACTION_WIDTH))

    def __len__(self):
        return self.length

    def append(self, states, actions, vc_goals=None, cc_goals=None):
        """database.py:104-146: rows go in one after the other; a full buffer drops its oldest row.  Vectorised: the
        result equals the reference's row-by-row loop."""
        if vc_goals is None and cc_goals is None:
            raise ValueError("both vc_goals and cc_goals cant be empty!")
        n = len(states)
        if n == 0:
            return
        idx = (self.start + self.length + np.arange(n)) % self.limit
        keep = slice(max(0, n - self.limit), n)           # rows that survive when n alone overflows the buffer
        self.states[idx[keep]] = np.asarray(states)[keep]
        self.actions[idx[keep]] = np.asarray(actions)[keep]
        if vc_goals is not None:
            self.vc_goals[idx[keep]] = np.asarray(vc_goals)[keep]
        if cc_goals is not None:
            self.cc_goals[idx[keep]] = np.asarray(cc_goals)[keep]
        over = max(0, self.length + n - self.limit)
        self.start = (self.start + over) % self.limit
        self.length = min(self.limit, self.length + n)

    def arrays(self):
        """the four arrays as save_dataset writes them: the first len(self) rows of the buffers (data_collection.py:107-114;
        after an overflow that is buffer order, not age order -- as in the reference)"""
        n = self.length
        return dict(states=self.states[:n], vc_goals=self.vc_goals[:n], cc_goals=self.cc_goals[:n], actions=self.actions[:n])

    def save(self, directory, iteration, config=None):
        os.makedirs(directory, exist_ok=True)
        arrs = self.arrays()
        try:
            import h5py
        except ImportError:
            path = os.path.join(directory, "database_%d.npz" % iteration)
            np.savez(path, **arrs)
        else:
            path = os.path.join(directory, "database_%d.hdf5" % iteration)
            with h5py.File(path, "w") as hf:
                for k, v in arrs.items():
                    hf.create_dataset(k, data=v)
        pkl = os.path.join(directory, "config.pkl")
        if config is not None and not os.path.exists(pkl):       # "save config as pickle only once" (data_collection.py:115-122)
            with open(pkl, "wb") as f:
                pickle.dump(_plain(config), f)
            with open(os.path.join(directory, "config.json"), "w") as f:
                json.dump(config, f, indent=1, default=str)
        return path


def _plain(x):
    """what OmegaConf.to_container leaves behind: dicts / lists / builtin scalars only"""
    if isinstance(x, dict):
        return {str(k): _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    if isinstance(x, np.generic):
        return x.item()
    if isinstance(x, np.ndarray):
        return x.tolist()
    if x is None or isinstance(x, (bool, int, float, str)):
        return x
    return str(x)


def vc_goal_rows(t, gait_period, v_des, w_des, gait_name):
    """simulation.py:177-187, 494-498: [phase, v_des x, v_des y, w_des, gait value] for times t (n,)"""
    t = np.asarray(t, float)
    out = np.zeros((t.shape[0], VC_GOAL_WIDTH))
    out[:, 0] = (t % gait_period) / gait_period
    out[:, 1:3] = np.asarray(v_des, float).reshape(-1, 3)[:, 0:2]
    out[:, 3] = w_des
    out[:, 4] = GAIT_VALUE.get(gait_name, 0.0)
    return out
