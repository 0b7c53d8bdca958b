"""TEST INFRASTRUCTURE: bench.py's command line without a GPU.  Stubs torch.cuda's three calls, routes the binding to the kernel
emulator (tests/emu) and runs bench.main() -- as its own script, so that `--gpus N` without a launcher re-executes THIS file under
torch.distributed.run (bench.self_launch starts sys.argv[0]) and every rank it starts is stubbed the same way.
Used by tests/test_bench_dryrun.py::test_unwrapped_multi_rank_command."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import torch
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda *a: None
    from helpers import emu
    import bench
    with emu.active():
        bench.main()


if __name__ == "__main__":
    main()
