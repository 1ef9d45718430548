"""Multi-GPU GBA of ONE merged map, sharded by agent (BASELINE.json north star: "the merged
multi-agent map shards by agent/sub-map across the GPUs of one node with RCCL all-reduce over xGMI on the shared-pose
Hessian blocks at each LM iteration").

One process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm). Every rank
  1. computes the SAME global plan from the full flat problem (`shard_plan`, host-only C entry point covgpu_shard_plan):
     agents' interiors = blocks owned by ranks, cross-agent "shared" keyframes = border, owner of every landmark /
     IMU factor / between factor;
  2. keeps its share (`shard_problem`: all K keyframes stay, residuals are filtered) and uploads it;
  3. runs the same trust-region loop; the library calls back into `reducer` for the three collectives of a linear
     solve (shared-pose gradient rows, the shared-pose system after the local interior eliminations, 16 scalars);
  4. `merge_solution` assembles the optimised map from the ranks' pieces.
The reducers: `TorchReducer` (RCCL on device pointers / gloo on host arrays) for real ranks, `ThreadReducer` for
VIRTUAL ranks — several contexts on one GPU driven by host threads, which is how the sharded path is verified on the
single-GPU test boxes.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import capi
from .capi import FlatProblem, Options

ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int32, C.c_int32)


def env_ranks() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str = "nccl", local_rank: int = 0, force: bool = False):
    """Returns the torch.distributed module (initialised) or None for a single process (force: a one-rank group)."""
    _, _, world = env_ranks()
    if world <= 1 and not force:
        return None
    if world <= 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def barrier(dist, device: Optional[str] = None) -> None:
    import torch
    if device is not None and device.startswith("cuda"):
        torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
        if device is not None and device.startswith("cuda"):
            torch.cuda.synchronize(device)


def aggregate(dt: float, iterations: float, dist, device: str = "cpu") -> Tuple[float, float]:
    """(max over ranks of the wall time, max over ranks of the executed iterations — every rank runs the SAME
    iterations of the one sharded solve, so this is the job's iteration count, not a sum)."""
    if dist is None:
        return float(dt), float(iterations)
    import torch
    t = torch.tensor([dt, float(iterations)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0].item()), float(t[1].item())


def throughput(dt_max: float, iterations: float) -> float:
    """Whole-job GBA iterations per second."""
    return iterations / dt_max


# ------------------------------------------------------------------------------------------------ plan / shards
@dataclass
class ShardPlan:
    world: int
    num_blocks: int
    block_of_kf: np.ndarray   # [K]  block (agent interior) of a keyframe, -1 = shared (border) keyframe
    block_rank: np.ndarray    # [num_blocks]
    lm_rank: np.ndarray       # [L]
    imu_rank: np.ndarray      # [I]
    edge_rank: np.ndarray     # [E]

    def pose_owner(self) -> np.ndarray:
        """rank whose result holds keyframe k's pose (shared keyframes: identical on every rank; take rank 0)."""
        return np.where(self.block_of_kf >= 0, self.block_rank[np.maximum(self.block_of_kf, 0)], 0).astype(np.int32)


def shard_plan(prob: FlatProblem, opt: Options, world: int) -> Optional[ShardPlan]:
    """Host-only and deterministic: every rank computes the same plan from the same full problem."""
    from . import backend
    bk = np.empty(prob.K, np.int32); br = np.zeros(max(prob.K, 1), np.int32)
    lr = np.zeros(max(prob.L, 1), np.int32); ir = np.zeros(max(prob.I, 1), np.int32); er = np.zeros(max(prob.E, 1), np.int32)
    s = prob.as_struct()
    n = backend.lib().covgpu_shard_plan(C.byref(opt), C.byref(s), int(world), capi.iptr(bk), capi.iptr(br), capi.iptr(lr), capi.iptr(ir), capi.iptr(er))
    if n <= 0:
        return None
    return ShardPlan(world, int(n), bk, br[:n].copy(), lr[:prob.L].copy(), ir[:prob.I].copy(), er[:prob.E].copy())


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


def chain_owner(prob: FlatProblem, plan: ShardPlan) -> np.ndarray:
    """rank that holds keyframe k's speed-bias block: the rank of its agent's IMU factors."""
    own = np.zeros(prob.K, np.int32)
    own[prob.imu_kf_j] = plan.imu_rank
    own[prob.imu_kf_i] = plan.imu_rank  # (a chain's first keyframe only appears as predecessor)
    return own


def merge_solution(prob: FlatProblem, plan: ShardPlan, parts: Sequence[FlatProblem]) -> FlatProblem:
    """Optimised full problem from the ranks' solved shares (parts[r] = rank r's downloaded sub-problem)."""
    out = prob.copy()
    po, so = plan.pose_owner(), chain_owner(prob, plan)
    for r, q in enumerate(parts):
        out.kf_pose[po == r] = q.kf_pose[po == r]
        out.kf_speed_bias[so == r] = q.kf_speed_bias[so == r]
        out.lm_pos[plan.lm_rank == r] = q.lm_pos
    return out


# ------------------------------------------------------------------------------------------------ reducers
class ThreadReducer:
    """All-reduce among VIRTUAL ranks = host threads of one process (each driving its own context, possibly on the same
    GPU). Host buffers only (covgpu_set_shard(..., stage_on_host=1)). The sum is formed in rank order by every thread:
    deterministic and identical everywhere."""

    def __init__(self, world: int):
        self.world = world
        self._bar = threading.Barrier(world)
        self._slots: List[Optional[np.ndarray]] = [None] * world
        self.calls = 0
        self.bytes = 0

    def callback(self, rank: int):
        def fn(user, buf, n, op, on_device):
            assert not on_device, "ThreadReducer needs stage_on_host=1"
            a = np.ctypeslib.as_array(buf, (int(n),))
            self._slots[rank] = a.copy()
            self._bar.wait()
            res = self._slots[0].copy()
            for r in range(1, self.world):
                res = res + self._slots[r] if op == 0 else np.maximum(res, self._slots[r])
            a[:] = res
            if rank == 0:
                self.calls += 1; self.bytes += 8 * int(n)
            self._bar.wait()
        return ALLREDUCE_FN(fn)


class TorchReducer:
    """All-reduce over a torch.distributed process group: RCCL on device pointers (bench.py: the buffer stays in HBM, the
    library has drained its stream before the call) and, for host buffers (the 16 scalars), through a small staging tensor
    on the group's device — or plain gloo on CPU."""

    def __init__(self, dist, device: str):
        self.dist, self.device = dist, device
        self.calls = 0
        self.bytes = 0

    def callback(self):
        import torch

        class _Dev:  # zero-copy view of a raw device pointer
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        def fn(user, buf, n, op, on_device):
            n = int(n)
            rop = self.dist.ReduceOp.SUM if op == 0 else self.dist.ReduceOp.MAX
            if on_device:
                t = torch.as_tensor(_Dev(C.cast(buf, C.c_void_p).value, n), device=self.device)
                self.dist.all_reduce(t, op=rop)
                torch.cuda.synchronize(self.device)
            else:
                a = np.ctypeslib.as_array(buf, (n,))
                t = torch.from_numpy(a.copy()).to(self.device)
                self.dist.all_reduce(t, op=rop)
                a[:] = t.cpu().numpy()
            self.calls += 1; self.bytes += 8 * n
        return ALLREDUCE_FN(fn)
