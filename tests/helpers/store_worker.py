"""Worker of tests/test_distrib.py::test_store_rendezvous_under_torchrun — launched by `python -m torch.distributed.run` exactly as
the driver launches bench.py: every rank opens covins_amd.distrib.open_store, rank 0 publishes 128 bytes (the place of the RCCL
unique id), every rank reads them back and publishes / gathers a small pickled piece (gather_solutions)."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from covins_amd import distrib  # noqa: E402
from covins_amd.capi import FlatProblem  # noqa: E402,F401

if __name__ == "__main__":
    rank, local_rank, world = distrib.env_ranks()
    store = distrib.open_store(rank, world)
    if rank == 0:
        store.set("rccl_id", bytes(range(128)))
    uid = bytes(store.get("rccl_id"))
    assert uid == bytes(range(128)), "unique id did not arrive"

    class Piece:  # what gather_solutions reads from a downloaded sub-problem
        kf_pose = np.full((3, 7), float(rank)); kf_speed_bias = np.full((3, 9), 10.0 + rank); lm_pos = np.full((2, 3), 100.0 + rank)
    got = distrib.gather_solutions(Piece, rank, world, store)
    assert len(got) == world
    for r, (kp, ks, lp) in enumerate(got):
        assert (kp == r).all() and (ks == 10.0 + r).all() and (lp == 100.0 + r).all()
    # the shard-plan digest exchange of distrib.attach: equal digests pass; with COVGPU_TEST_BAD_DIGEST=1 rank 1 publishes another one and
    # EVERY rank must refuse
    digest = "a" * 64 if not (os.environ.get("COVGPU_TEST_BAD_DIGEST") == "1" and rank == 1) else "b" * 64
    refused = False
    try:
        distrib.check_plan_digest(store, rank, world, digest)
    except Exception as e:
        refused = "digest" in str(e)
    assert refused == (os.environ.get("COVGPU_TEST_BAD_DIGEST") == "1"), "plan digest check"
    out = os.environ["COVGPU_TEST_OUT"]
    with open(f"{out}.{rank}", "w") as f:
        f.write(hashlib.sha256(uid).hexdigest())
