"""Multi-GPU GBA of ONE merged map, sharded by sub-map (BASELINE.json north star: "the merged
multi-agent map shards by agent/sub-map across the GPUs of one node with RCCL all-reduce over xGMI on the shared-pose
Hessian blocks at each LM iteration").

The unit of the split is a SUBTREE of the multifrontal elimination tree (covins_amd/csrc/nd_plan.hip): an agent, or a
stretch of an agent's trajectory. The top of the tree — the separators that join the sub-maps, i.e. the shared poses — is
replicated. Every rank
  1. computes the SAME global plan from the full flat problem (`shard_plan`, host-only C entry point covgpu_shard_plan):
     tree node -> rank (or top), owner of every landmark / IMU factor / between factor;
  2. keeps its share (`shard_problem`: all K keyframes stay, residuals are filtered), attaches the native collective
     (`attach`: RCCL inside libcovgpu — one process per GPU, unique id passed through a torch TCPStore — or an in-process
     `Group` of host threads for VIRTUAL ranks on one GPU, which is how the sharded path is verified on single-GPU boxes) and
     uploads it;
  3. runs the same device-side trust-region loop; per iteration the library enqueues four all-reduces on its own stream
     (top fronts + right-hand sides + gradient rows once per linear solve, three small scalar exchanges);
  4. `merge_solution` assembles the optimised map from the ranks' pieces.
No Python code sits on the data path: the collectives are issued by libcovgpu (solver.hip: RcclReducer / GroupReducer).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

from . import capi
from .capi import FlatProblem, Options


def env_ranks() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


# ------------------------------------------------------------------------------------------------ plan / shards
class ShardPlan:
    """Global plan of a sharded solve: owns the native covgpu_nd_plan (tree + node -> rank)."""

    def __init__(self, handle, world: int, subtrees: int, lm_rank, imu_rank, edge_rank, pose_rank, sb_rank, node_rank):
        self.handle, self.world, self.subtrees = handle, world, subtrees
        self.lm_rank, self.imu_rank, self.edge_rank = lm_rank, imu_rank, edge_rank
        self.pose_rank, self.sb_rank = pose_rank, sb_rank   # [K] rank whose result holds the keyframe's pose / speed-bias, -1 = top (every rank)
        self.node_rank = node_rank                          # [tree nodes] owning rank, -1 = top (replicated)

    def close(self):
        if self.handle:
            from . import backend
            backend.lib().covgpu_nd_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_plan(prob: FlatProblem, opt: Options, world: int) -> Optional[ShardPlan]:
    """Host-only and deterministic: every rank computes the same plan from the same full problem."""
    from . import backend
    lib = backend.lib()
    lr = np.zeros(max(prob.L, 1), np.int32); ir = np.zeros(max(prob.I, 1), np.int32); er = np.zeros(max(prob.E, 1), np.int32)
    s = prob.as_struct()
    h = C.c_void_p()
    n = lib.covgpu_shard_plan(C.byref(opt), C.byref(s), int(world), C.byref(h), capi.iptr(lr), capi.iptr(ir), capi.iptr(er))
    if n <= 0 or not h:
        return None
    pr = np.zeros(prob.K, np.int32); sr = np.zeros(prob.K, np.int32)
    lib.covgpu_nd_plan_owner(h, capi.iptr(pr), capi.iptr(sr))
    info = (C.c_int64 * 16)()
    lib.covgpu_nd_plan_info(h, info)
    nr = np.zeros(int(info[0]), np.int32)
    lib.covgpu_nd_plan_ranks(h, capi.iptr(nr))
    plan = ShardPlan(h, int(world), int(n), lr[:prob.L].copy(), ir[:prob.I].copy(), er[:prob.E].copy(), pr, sr, nr)
    plan.top_mode, plan.leaf, plan.group_frac = int(info[10]), int(info[11]), int(info[12]) / 100.0   # which candidate tree was kept (COVGPU_ND_TOP / COVGPU_ND_LEAF / COVGPU_ND_GROUP_FRAC reproduce it on one GPU)
    return plan


def plan_digest(plan: ShardPlan) -> str:
    """sha256 over everything that shapes the exchange: tree node -> rank, owner of every landmark / IMU factor / between factor, owner of
    every keyframe's pose and speed-bias block."""
    import hashlib
    h = hashlib.sha256()
    h.update(np.int64([plan.world, plan.subtrees]).tobytes())
    for a in (plan.node_rank, plan.lm_rank, plan.imu_rank, plan.edge_rank, plan.pose_rank, plan.sb_rank):
        a = np.ascontiguousarray(a, np.int32)
        h.update(np.int64([a.size]).tobytes()); h.update(a.tobytes())
    return h.hexdigest()




def check_plan_digest(store, rank: int, world: int, mine: str) -> None:
    """Every rank publishes its digest and reads everybody's: ALL ranks raise on a mismatch (nobody is left waiting in a collective).
    The keys carry the attach's sequence number (a second attach on the same store must not read the first one's digests), and
    a second round of keys — every rank's verdict — follows the comparison, so that no rank goes on to ncclCommInitRank while another
    has already refused."""
    from . import backend
    # the attach number comes from the STORE (a counter keyed by id(store) can be inherited by a fresh store that reuses the id in one
    # process and not in another — the ranks would then wait for differently numbered keys): every rank counts its own attaches under its own key
    seq = int(store.add(f"plan_digest_seq_{rank}", 1)) - 1
    store.set(f"plan_digest_{seq}_{rank}", mine.encode())
    bad = None
    for r in range(world):
        other = bytes(store.get(f"plan_digest_{seq}_{r}")).decode()
        if other != mine and bad is None:
            bad = (r, other)
    store.set(f"plan_digest_ok_{seq}_{rank}", b"0" if bad else b"1")
    refused = [r for r in range(world) if bytes(store.get(f"plan_digest_ok_{seq}_{r}")) != b"1"]
    if bad:
        raise backend.CovGpuError(f"rank {rank}: shard plan digest {mine[:16]}.. differs from rank {bad[0]}'s {bad[1][:16]}..: the ranks did not "
                                  "compute the same plan (different problem or options) — aborting before any collective is issued")
    if refused:
        raise backend.CovGpuError(f"rank {rank}: rank(s) {refused} refused the shard plan (their digest differs from another rank's) — aborting "
                                  "before any collective is issued")


def shard_problem(prob: FlatProblem, plan: ShardPlan, rank: int) -> FlatProblem:
    """Rank `rank`'s share: every keyframe (states are replicated, 128 B each), its landmarks with their observations,
    its IMU factors, its between factors."""
    lm = plan.lm_rank == rank
    n_obs = np.diff(prob.lm_obs_ptr)
    obs = np.repeat(lm, n_obs)
    imu = plan.imu_rank == rank
    n_smp = np.diff(prob.imu_sample_ptr)
    smp = np.repeat(imu, n_smp)
    ed = plan.edge_rank == rank
    return FlatProblem(
        kf_pose=prob.kf_pose, kf_speed_bias=prob.kf_speed_bias, kf_fixed=prob.kf_fixed, kf_cam=prob.kf_cam,
        cam_extr=prob.cam_extr, cam_intr=prob.cam_intr, cam_dist=prob.cam_dist, cam_dist_type=prob.cam_dist_type,
        lm_pos=prob.lm_pos[lm], lm_obs_ptr=np.concatenate([[0], np.cumsum(n_obs[lm])]).astype(np.int32),
        obs_kf=prob.obs_kf[obs], obs_uv=prob.obs_uv[obs], obs_sigma=prob.obs_sigma[obs],
        imu_kf_i=prob.imu_kf_i[imu], imu_kf_j=prob.imu_kf_j[imu],
        imu_sample_ptr=np.concatenate([[0], np.cumsum(n_smp[imu])]).astype(np.int32), imu_samples=prob.imu_samples[smp],
        imu_first=prob.imu_first[imu], imu_noise=None if prob.imu_noise is None else prob.imu_noise[imu],
        edge_i=prob.edge_i[ed], edge_j=prob.edge_j[ed], edge_meas=prob.edge_meas[ed], edge_sqrt_info=prob.edge_sqrt_info[ed],
        edge_loss_a=prob.edge_loss_a[ed])


def merge_solution(prob: FlatProblem, plan: ShardPlan, parts: Sequence[FlatProblem]) -> FlatProblem:
    """Optimised full problem from the ranks' solved shares (parts[r] = rank r's downloaded sub-problem). Top unknowns are
    identical on every rank: rank 0's copy is taken."""
    out = prob.copy()
    po = np.where(plan.pose_rank < 0, 0, plan.pose_rank); so = np.where(plan.sb_rank < 0, 0, plan.sb_rank)
    for r, q in enumerate(parts):
        out.kf_pose[po == r] = q.kf_pose[po == r]
        out.kf_speed_bias[so == r] = q.kf_speed_bias[so == r]
        out.lm_pos[plan.lm_rank == r] = q.lm_pos
    return out


# ------------------------------------------------------------------------------------------------ collectives
class Group:
    """In-process group of ranks = host threads, each driving its own context on a shared device (covgpu_group_*)."""

    def __init__(self, world: int):
        from . import backend
        self.world = world
        self.handle = C.c_void_p()
        rc = backend.lib().covgpu_group_create(int(world), C.byref(self.handle))
        if rc != 0:
            raise backend.CovGpuError(backend.lib().covgpu_last_error().decode())

    def abort(self):
        from . import backend
        backend.lib().covgpu_group_abort(self.handle)

    def close(self):
        if self.handle:
            from . import backend
            backend.lib().covgpu_group_destroy(self.handle)
            self.handle = None


def attach(ctx, plan: ShardPlan, rank: int, world: int, force_single: bool = False, group: Optional[Group] = None, store=None, tag: str = ""):
    """Gives `ctx` its collective. Real ranks (world > 1, one process per GPU): RCCL inside libcovgpu; rank 0 creates the
    unique id and passes it to the others through a torch.distributed TCPStore on MASTER_ADDR:MASTER_PORT (key exchange
    only — no torch process group, no tensor ever touches it). A one-rank run with force_single exercises the whole sharded
    path through an in-process group of one. `store` / `tag`: a second context of the same process (another workload of the same bench run)
    reuses the first one's store under its own key names. Returns the objects that must stay alive with the context."""
    if group is not None:
        ctx.set_shard_group(plan, rank, group)
        return (group,)
    from . import backend
    if world <= 1:
        # one rank: the RCCL form (a one-rank communicator: collectives stream-ordered, no host synchronisation — what every rank of
        # a real run executes); the in-process group (host barriers: the test vehicle of the virtual ranks) only without librccl
        buf = (C.c_uint8 * 128)()
        if backend.lib().covgpu_rccl_unique_id(buf) == 0:
            ctx.set_shard_rccl(plan, 0, 1, bytes(buf))
            return ("rccl",)
        g = Group(1)
        ctx.set_shard_group(plan, 0, g)
        return (g,)
    if store is None:
        store = open_store(rank, world)
    else:
        from torch.distributed import PrefixStore
        store = PrefixStore(f"{tag}/", store)
    # every rank computed the plan ITSELF from its copy of the problem: a rank whose copy differs (another map revision, another
    # option, a non-deterministic generator) would exchange fronts of another shape and hang or corrupt the solve. Digests first.
    check_plan_digest(store, rank, world, plan_digest(plan))
    if rank == 0:
        buf = (C.c_uint8 * 128)()
        rc = backend.lib().covgpu_rccl_unique_id(buf)
        if rc != 0:
            raise backend.CovGpuError(backend.lib().covgpu_last_error().decode())
        store.set("rccl_id", bytes(buf))
    uid = store.get("rccl_id")
    ctx.set_shard_rccl(plan, rank, world, bytes(uid))
    return (store,)


def open_store(rank: int, world: int):
    """Key-value store on MASTER_ADDR:MASTER_PORT for the few bytes the ranks exchange outside the data path (the RCCL unique id,
    the solved pieces at the end of a bench run). Under `python -m torch.distributed.run` the elastic agent already serves a
    TCPStore on that port (TORCHELASTIC_USE_AGENT_STORE): every rank connects as a client, exactly as torch's own env://
    rendezvous does; launched by hand, rank 0 hosts it."""
    import datetime
    from torch.distributed import PrefixStore, TCPStore
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")), world, rank == 0 and not agent,
                     timeout=datetime.timedelta(seconds=300), wait_for_workers=False)
    return PrefixStore(f"covgpu/{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}/", store)


def barrier(ctx, sharded: bool) -> None:
    """All ranks have drained their work (an all-reduce of one number through the context's collective)."""
    if sharded:
        ctx.allreduce_host(np.zeros(1), 0)


def aggregate(dt: float, iterations: float, ctx, sharded: bool) -> Tuple[float, float]:
    """(max over ranks of the wall time, max over ranks of the executed iterations — every rank runs the SAME iterations
    of the one sharded solve, so this is the job's iteration count, not a sum)."""
    if not sharded:
        return float(dt), float(iterations)
    r = ctx.allreduce_host(np.array([dt, float(iterations)]), 1)
    return float(r[0]), float(r[1])


def throughput(dt_max: float, iterations: float) -> float:
    """Whole-job GBA iterations per second."""
    return iterations / dt_max


def gather_solutions(sol: FlatProblem, rank: int, world: int, store) -> Sequence:
    """All ranks' (poses, speed-bias, landmarks) on every rank, through the TCPStore of `attach` (small: a few MB). Raw float64
    buffers, never pickles: the store is an unauthenticated TCP service, and what comes out of it is only ever reinterpreted as
    numbers (shapes follow from K, which every rank knows, and from the buffer length)."""
    K = int(sol.kf_pose.shape[0])
    parts = [np.ascontiguousarray(a, np.float64).reshape(-1) for a in (sol.kf_pose, sol.kf_speed_bias, sol.lm_pos)]
    assert parts[0].size == 7 * K and parts[1].size == 9 * K and parts[2].size % 3 == 0
    store.set(f"sol_{rank}", np.concatenate(parts).tobytes())
    out = []
    for r in range(world):
        buf = np.frombuffer(bytes(store.get(f"sol_{r}")), np.float64)
        if buf.size < 16 * K or (buf.size - 16 * K) % 3:
            raise ValueError(f"rank {r}: solution buffer of {buf.size} doubles does not fit K={K}")
        out.append((buf[:7 * K].reshape(K, 7).copy(), buf[7 * K:16 * K].reshape(K, 9).copy(), buf[16 * K:].reshape(-1, 3).copy()))
    return out
