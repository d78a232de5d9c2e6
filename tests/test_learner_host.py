"""Host-side logic of the learner's whole-network kernels (no GPU): the packed-weight table and the quad-row layout helpers
of engine.py against their definitions in include/odk.h (odk_mlp_desc, odk_weight_table, odk_dw_gemm)."""
import numpy as np
import torch

from open_duck_playground_amd import engine


def test_weight_table_offsets_and_sizes():
    # policy (101 -> 512 -> 256 -> 128 -> 28) and value (212 -> ... -> 1) of the reference, as FlatLearner lays them out
    entries, off = [], 0
    for n_in, n_out in ((101, 28), (212, 1)):
        widths = (n_in,) + engine.MLP_HIDDEN + (n_out,)
        for l in range(4):
            entries.append((off, widths[l + 1], widths[l], l > 0))
            off += widths[l + 1] * widths[l] + widths[l + 1]
    t = engine.WeightTable(entries)
    pad = lambda k: (k + 15) // 16 * 16
    fo = bo = 0
    for k, (o, r, c, bw) in enumerate(entries):
        assert (t.c.off[k], t.c.rows[k], t.c.cols[k]) == (o, r, c)
        assert t.c.fwd_off[k] == fo and fo % 4 == 0
        fo += pad(c) * r
        if bw:
            assert t.c.bwd_off[k] == bo and bo % 4 == 0
            bo += pad(r) * c
        else:
            assert t.c.bwd_off[k] == -1 and t.bwd_view(torch.zeros(8), k) is None
    assert (t.fwd_size, t.bwd_size, t.c.count) == (fo, bo, 8)
    # the first layer's input widths are the only padded reduction lengths of the forward copies
    assert t.fwd[0][1] == 112 * 512 and t.fwd[4][1] == 224 * 512 and t.bwd[3][1] == 32 * 128 and t.bwd[7][1] == 16 * 128
    buf = torch.arange(t.fwd_size, dtype=torch.float32)
    v = t.fwd_view(buf, 5)
    assert v.numel() == 512 * 256 and float(v[0]) == t.c.fwd_off[5]


def test_quad_row_layout_roundtrip_and_definition():
    g = torch.Generator().manual_seed(0)
    for n, w in ((5, 3), (16, 28), (37, 101), (320, 512)):
        x = torch.randn(n, w, generator=g)
        q = engine.quad_pack(x)
        np_ = engine.quad_rows(n)
        assert np_ % 16 == 0 and np_ >= n and q.numel() == np_ * w
        assert torch.equal(engine.quad_unpack(q, n, w), x)
        assert float(engine.quad_unpack(q, np_, w)[n:].abs().sum()) == 0.0        # rows past the batch: zeros
        qa = q.numpy()
        for s, f in ((0, 0), (n - 1, w - 1), (n // 2, w // 3)):                    # element (s, f) at ((s / 4) * width + f) * 4 + s % 4
            assert qa[((s // 4) * w + f) * 4 + s % 4] == np.float32(x[s, f])
