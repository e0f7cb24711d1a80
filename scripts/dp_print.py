"""Output helper of the two-rank scripts (scripts/dp_*_smoke.py); no side effects at import."""
def print_in_rank_order(line):
    """One JSON line per rank on the shared stdout pipe, rank 0 first: concurrent writes of long lines interleave (a test then fails to
    parse them), so the ranks take turns behind barriers."""
    import sys
    import torch.distributed as dist
    for r in range(dist.get_world_size()):
        if dist.get_rank() == r:
            print(line)
            sys.stdout.flush()
        dist.barrier()
