"""Test-only launcher: runs bench.main() with the kernel emulator injected (CPU, gloo) so that the rank plumbing,
the barrier / max-over-ranks timing, the metrics all-gather and the JSON contract of bench.py are exercised without a
GPU. The timings it prints are meaningless."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def factory(n_envs, cap, arr, outputs):
    import pve_mcc_amd
    from tests.hip_adapter import emulator_lib
    return pve_mcc_amd.BatchedIntersections(n_envs, cap, arr, device="cpu", outputs=outputs, _lib=emulator_lib())


if __name__ == "__main__":
    bench.main(sys.argv[1:], env_factory=factory)
