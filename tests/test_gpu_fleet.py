"""FleetPlanner on the MI355X: the script of tests/test_fleet.py with the real BatchSolver behind it -- every robot of the batch reproduces the recorded run of the reference's
plugin (reference plugin + reference Controller + the C oracle's solve, recorded on the CPU) within the north-star tolerance."""
import pytest

from test_fleet import LOOPS, replay


@pytest.mark.gpu
@pytest.mark.parametrize("loop", LOOPS)
def test_fleet_cycle_on_the_gpu_reproduces_the_recorded_runs_of_the_reference_plugin(loop):
    worst_cmd, worst_x = replay(loop, None, [0, 5, 0, None], 1e-4)
    print(f"{loop}: fleet cycle on the GPU against the recorded plugin runs: largest command difference {worst_cmd:.2e}, largest state difference {worst_x:.2e}")
