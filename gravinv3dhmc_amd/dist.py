"""One-process-per-GPU launch helpers (the reference's `mpiexec -n K python main_*.py`).

The reference's only use of MPI is `Get_rank()/Get_size()` (example/*/main_*.py:20-22):
every rank is an independent chain with seed + rank and its own output folder
(hmc.py:367-369) -- there is no exchange step on the data path, so none is added here.
`torch.distributed` (gloo) is used purely as a control plane: barrier, max of timings and
gathering per-chain summaries on rank 0.  Launch with
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 your_script.py
"""
import os


class Ranks(object):
    """Rank bookkeeping of an N-chain job; degenerates to a single chain without a launcher."""

    def __init__(self, init=True):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self._dist = None
        if init and self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self._dist = dist

    @property
    def device(self):
        """GPU ordinal of this rank (one process per GPU)."""
        return self.local_rank

    def chain_seed(self, seed):
        return seed + self.rank          # hmc.py:369

    def chain_folder(self, save_folder):
        return save_folder + str(self.rank)   # hmc.py:368

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max(self, value):
        """Max of a Python float over ranks (timing of the slowest chain)."""
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])

    def sum(self, value):
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t[0])

    def gather(self, obj):
        """List of every rank's picklable `obj` on rank 0 (None elsewhere): e.g. posterior
        mean/std of each chain for the multi-chain statistics of plot_real_multichain.py:64-77."""
        if self._dist is None:
            return [obj]
        out = [None] * self.world if self.rank == 0 else None
        self._dist.gather_object(obj, out, dst=0)
        return out

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._dist = None


def run_chains(model_factory, sample_kwargs, ranks=None):
    """Build this rank's model on its GPU and run its chain (HMCSample) -- the body of the
    reference drivers' main() (example/uniformgrid/main_uniform.py:17-88)."""
    from .inversion.hmc import HMCSample
    ranks = ranks or Ranks()
    model = model_factory(ranks.device)
    kw = dict(sample_kwargs)
    kw["myrank"] = ranks.rank
    chain = HMCSample(model, **kw)
    ranks.barrier()
    return chain
