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


def test_a_dropped_ticket_frees_its_model():
    """`busy` was a bare lock held from submit to collect, so a ticket lost to an exception left the model locked for good.
    A LaneTicket releases the lane when it is closed (collect) or garbage-collected (dropped), and closing twice is harmless."""
    import gc
    from mangatranslator_amd.hip.plan import AsyncLane, LaneTicket, result_tensors
    import torch
    lane = AsyncLane("cpu", simulator=True)
    lane.acquire()
    t = LaneTicket(lane, plan="p", hw=(3, 4))
    assert dict(t) == {"plan": "p", "hw": (3, 4)} and lane.busy.locked()        # collect halves expand it with ** like a dict
    t.close(); t.close()
    assert not lane.busy.locked()
    lane.acquire()
    t = LaneTicket(lane, x=1)
    del t
    gc.collect()
    assert not lane.busy.locked()
    lane.release()                                                               # releasing a free lane is a no-op
    assert lane.busy.acquire(blocking=False)
    lane.release()
    # what hand_over records: tensors inside result namespaces, lists and tuples (CUDA ones only)
    import types
    res = [types.SimpleNamespace(boxes=types.SimpleNamespace(xyxy=torch.zeros(2, 4)), masks=None, names={0: "a"})]
    assert result_tensors(res) == [] and result_tensors((torch.zeros(1), [torch.zeros(1)])) == []
