import pytest

import plan_lane_checks as plc


def test_lane_flags_and_program_order(emu_lib):
    """CPU tier: the builder stamps the lanes; the simulator runs the ops in program order, which is a valid serial order"""
    plc.check(emu_lib, "cpu", graph=False)


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_lanes_on_hardware(hip_lib, graph):
    """GPU tier: the side lane runs on its own stream (eager) / as a parallel branch of the captured hipGraph"""
    plc.check(hip_lib, "cuda:0", graph=graph)
