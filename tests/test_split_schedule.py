"""CPU: a model of the activation hand-over of the split kernels (nerfmeshes_amd/csrc/mlp_device_gs.h, gs_publish_first /
gs_stage_hidden): for every width class they serve, replay the schedule chunk by chunk and check the protocol's three claims --
a tile is visible (written before a barrier) before its reader fetches it, it is fetched before the chunk that multiplies with it,
and a slot is never rewritten before a barrier behind the last fetch of the tile it held.  The kernels themselves are pinned bit
for bit to the one-wave kernels on the GPU (tests/tools/bench_split.py, tests/test_gpu_generic.py); this pins the reasoning."""
import pytest

SLOTS = 4          # 1 KiB exchange slots per pair of waves
KCH = 4            # k-steps per weight chunk = one 16-feature input tile


def owner(t, nti):
    return 0 if t < (nti + 1) // 2 else 1          # GsHalf: wave 0 owns tiles [0, ceil(n / 2)), wave 1 the rest


def schedule(nti):
    """events (time, kind, wave, tile); time = 2 * barrier index + phase: the barrier in front of the stage is index 0, the one
    that ends chunk c is index c + 1; work between barriers b and b + 1 happens at time 2 b + 1."""
    ev = []
    for t in (0, 1):                                # gs_publish_first: before the stage's first barrier
        if t < nti:
            ev.append((-1, "write", owner(t, nti), t))
    reader0 = 1 - owner(0, nti)
    ev.append((1, "read", reader0, 0))              # right behind that barrier (chunk 0's interval)
    for c in range(nti):
        now = 2 * c + 1                             # inside chunk c
        if c + 2 < nti:
            ev.append((now, "write", owner(c + 2, nti), c + 2))
        if c + 1 < nti:
            ev.append((now, "read", 1 - owner(c + 1, nti), c + 1))
        ev.append((now, "use", 1 - owner(c, nti), c))
    return ev


@pytest.mark.parametrize("nti", [13, 14, 15, 16, 26, 28, 30, 32])   # the view layer's tiles (backward: 13 -- 16) and the trunk's
def test_exchange_schedule_is_race_free(nti):
    ev = schedule(nti)
    write = {t: time for time, kind, _, t in ev if kind == "write"}
    read = {t: time for time, kind, _, t in ev if kind == "read"}
    use = {t: time for time, kind, _, t in ev if kind == "use"}
    assert sorted(write) == sorted(read) == sorted(use) == list(range(nti)), "every tile crosses exactly once"
    for t in range(nti):
        # a barrier lies between two events iff an even number lies strictly between their times
        assert write[t] + 1 < read[t] or (write[t] < read[t] and (write[t] + 1) % 2 == 0), (t, "written before a barrier in front of its fetch")
        assert read[t] <= use[t], (t, "fetched no later than the chunk that multiplies with it")
        if t + SLOTS < nti:
            # the next tenant of slot t % SLOTS: its write must come after a barrier that follows the fetch of tile t
            assert write[t + SLOTS] > read[t] + 1 or (write[t + SLOTS] > read[t] and (read[t] + 1) % 2 == 0), (t, "slot rewritten under its reader")
    # at most two fetched-but-unused tiles are alive in a wave's registers (xb[2])
    for wave in (0, 1):
        for now in sorted({time for time, *_ in ev}):
            alive = [t for t in range(nti) if 1 - owner(t, nti) == wave and read[t] <= now <= use[t]]
            assert len(alive) <= 2, (wave, now, alive)
