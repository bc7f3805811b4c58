"""bunmpc_amd/dataset.py against the row-by-row ring buffer of the reference's Database.append (database.py:104-146,
restated here as the loop it is) and its save layout."""
import numpy as np

from bunmpc_amd import dataset


def _loop_append(buf, start, length, limit, rows):
    for r in rows:
        if length < limit:
            length += 1
        else:
            start = (start + 1) % limit
        buf[(start + length - 1) % limit] = r
    return start, length


def test_ring_buffer_equals_the_row_by_row_loop():
    rng = np.random.default_rng(0)
    db = dataset.Database(limit=50)
    ref = np.zeros((50, 43))
    start = length = 0
    for n in (7, 30, 20, 0, 49, 120, 3):
        s, a, g = rng.normal(size=(n, 43)), rng.normal(size=(n, 12)), rng.normal(size=(n, 5))
        db.append(s, a, vc_goals=g)
        start, length = _loop_append(ref, start, length, 50, s)
        assert (db.start, db.length) == (start, length) and np.array_equal(db.states, ref)
    assert len(db) == 50
    try:
        db.append(s, a)
    except ValueError:
        pass
    else:
        raise AssertionError("goal-less append must be refused")


def test_save_layout(tmp_path):
    db = dataset.Database(limit=10)
    rng = np.random.default_rng(1)
    db.append(rng.normal(size=(4, 43)), rng.normal(size=(4, 12)), vc_goals=rng.normal(size=(4, 5)), cc_goals=rng.normal(size=(4, 12)))
    cfg = {"gaits": ["trot"], "n_iteration": np.int64(20), "sigma": np.array([0.1, 0.2]), "nested": {"episode_length": 3000, "v": (0.0, 0.3)}}
    path = db.save(str(tmp_path), 3, config=cfg)
    assert path.endswith("database_3.npz") or path.endswith("database_3.hdf5")
    if path.endswith(".npz"):
        z = np.load(path)
        assert sorted(z.files) == ["actions", "cc_goals", "states", "vc_goals"]
        assert z["states"].shape == (4, 43) and z["actions"].shape == (4, 12) and z["vc_goals"].shape == (4, 5)
        assert np.array_equal(z["states"], db.states[:4])
    assert (tmp_path / "config.json").exists()
    # config.pkl: a stdlib pickle of a plain dict, as data_collection.py:116-122 writes it (written once, kept on later saves)
    import pickle
    with open(tmp_path / "config.pkl", "rb") as f:
        got = pickle.load(f)
    assert got == {"gaits": ["trot"], "n_iteration": 20, "sigma": [0.1, 0.2], "nested": {"episode_length": 3000, "v": [0.0, 0.3]}}
    assert type(got["n_iteration"]) is int and type(got["sigma"]) is list
    db.save(str(tmp_path), 4, config={"other": 1})
    with open(tmp_path / "config.pkl", "rb") as f:
        assert pickle.load(f) == got


def test_vc_goal_rows():
    g = dataset.vc_goal_rows(np.array([0.0, 0.125, 0.5, 0.625]), 0.5, np.array([[0.3, 0.1, 0.0]]), 0.2, "trot")
    assert np.allclose(g[:, 0], [0.0, 0.25, 0.0, 0.25]) and np.all(g[:, 1] == 0.3) and np.all(g[:, 2] == 0.1)
    assert np.all(g[:, 3] == 0.2) and np.all(g[:, 4] == 1.0)
    assert dataset.vc_goal_rows(np.zeros(1), 0.5, np.zeros((1, 3)), 0.0, "pace")[0, 4] == 0.0
