"""The arithmetic the pipelined entropy kernel's streamed body rests on (cool_chic_amd/csrc/ccd_entropy_pipe.hip: StreamBody,
grid_segments), restated in Python and checked against the wavefront order itself (latent.py:66-140: pixels of a step share
x + 10 y, steps in increasing order, rows top to bottom inside a step):

* the steps [230, W + 10 (H - 24)) of a grid with W > 230 and H >= 25 all hold >= 23 pixels, the 230 steps in front of them
  and the 230 behind them hold 2 760 pixels each (closed forms of StreamBody::init);
* in that body a pixel's left neighbour lies >= 18 places earlier in decoding order, i.e. in an EARLIER 16-pixel batch than any
  8-pixel task that holds the pixel - the condition under which batches may be cut without regard to step ends;
* the left neighbour of the pixel with index i of a step is the pixel with index i (+ 1 when the step start moved down a row)
  of the previous step, and an 8-pixel task of the stream holds pixels of at most two steps - what the producers' task header
  computes per lane."""
import numpy as np
import pytest

T = 24  # kStreamMinStep


def wavefront(h, w):
    """(step index, y, x) of every pixel in decoding order, and the first pixel's stream position per step."""
    order = []
    starts = []
    for c in range(w + 10 * (h - 1)):
        starts.append(len(order))
        y0 = 0 if c < w else (c - w) // 10 + 1
        x0 = c if c < w else w - 10 + (c - w) % 10
        n = min(h - y0, x0 // 10 + 1)
        for i in range(n):
            order.append((c, y0 + i, x0 - 10 * i))
    return order, starts


@pytest.mark.parametrize("h,w", [(25, 241), (25, 250), (33, 480), (40, 271), (64, 521), (300, 241), (57, 332), (128, 256), (96, 768)])
def test_streamed_body_closed_forms_and_dependency_distance(h, w):
    order, starts = wavefront(h, w)
    n_steps = w + 10 * (h - 1)
    first, end = 10 * (T - 1), w + 10 * (h - T)
    assert first < end <= n_steps
    lens = np.diff(np.array(starts + [len(order)]))
    assert lens[first:end].min() >= T - 1
    assert starts[first] == 5 * T * (T - 1) == 2760
    assert len(order) - starts[end] == 2760 if end < n_steps else True
    assert starts[end] - starts[first] == h * w - 2 * 2760
    pos = {(y, x): p for p, (_, y, x) in enumerate(order)}
    body0 = starts[first]
    for p in range(starts[first], starts[end]):
        c, y, x = order[p]
        if x == 0:
            continue
        left = pos[(y, x - 1)]
        assert order[left][0] == c - 1                      # the left neighbour is decoded one step earlier ...
        i = p - starts[c]
        moved = 1 if (c >= w and (c - w) % 10 == 0) else 0
        assert left - starts[c - 1] == i + moved            # ... at the same index (+ 1 when the step start moved down a row)
        assert p - left >= 18
        task_first = body0 + ((p - body0) // 8) * 8         # first pixel of the stream task that holds p
        batch_first = body0 + ((p - body0) // 16) * 16
        assert left < batch_first and left < task_first     # an earlier batch: decoded before this task's batch can start
        # a task holds pixels of at most two steps
        last = min(task_first + 7, starts[end] - 1)
        assert order[last][0] - order[task_first][0] <= 1
