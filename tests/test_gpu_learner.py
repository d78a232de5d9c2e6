"""GPU learner kernels (csrc/odk_learner.hip, ppo/learner.py) against the autograd reference in ppo/train.py."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _fake_rollout(N, T, dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    done = (torch.rand(N, T, device=dev, generator=g) < 0.08).float()
    trunc = (torch.rand(N, T, device=dev, generator=g) < 0.5).float() * done
    return dict(obs=r(N, T, 101), priv=r(N, T, 212), raw_action=0.7 * r(N, T, 14), log_prob=-12 + r(N, T), reward=0.05 * r(N, T).abs(),
                done=done, truncation=trunc, last_priv=r(N, 212))


@pytest.mark.parametrize("B, Tn", [(256, 20), (37, 5), (300, 20), (1200, 40)])   # LDS-staged, register-staged and generic kernels
def test_gae_kernel_matches_torch_reference(B, Tn):
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.ppo import train as T
    g = torch.Generator(device="cuda").manual_seed(0)
    rew, val = torch.randn(B, Tn, device="cuda", generator=g), torch.randn(B, Tn, device="cuda", generator=g)
    boot = torch.randn(B, device="cuda", generator=g)
    term = (torch.rand(B, Tn, device="cuda", generator=g) < 0.1).float()
    trunc = (torch.rand(B, Tn, device="cuda", generator=g) < 0.05).float() * (1 - term)
    stats = torch.zeros(2, device="cuda")
    vs, adv = engine.gae(trunc, term, rew, val, boot, 0.95, 0.97, stats=stats)
    tm = lambda x: x.transpose(0, 1)
    vs_ref, adv_ref = T.compute_gae(tm(trunc), tm(term), tm(rew), tm(val), boot, 0.95, 0.97)
    torch.testing.assert_close(vs, tm(vs_ref).contiguous(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(adv, tm(adv_ref).contiguous(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(stats[0], adv_ref.mean(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(stats[1], 1.0 / (adv_ref.std(unbiased=False) + 1e-8), rtol=1e-4, atol=1e-5)


def test_dw_gemm_matches_torch_mm_and_is_reproducible():
    """odk_dw_gemm (weight gradients dz^T h of several layers, split-K on the f32 matrix cores, quad-row operands) vs float64
    torch; edge tiles (28, 1, 101, 212 columns), ragged row counts (slices of unequal length), offsets into a flat buffer,
    bit-identical repeats (fixed-order fold of the row slices)."""
    from open_duck_playground_amd import engine
    g = torch.Generator(device="cuda").manual_seed(0)
    for n, shapes, ks in ((1280, [(512, 101), (256, 512), (128, 256), (28, 128)], 8), (777, [(512, 212), (1, 128)], 16), (200, [(36, 70), (64, 33)], 8)):
        tot = sum((o * i + 7) // 4 * 4 for o, i in shapes) + 12
        flat = torch.full((tot,), 7.0, device="cuda")
        ws = torch.empty(ks * engine.DwGemm.workspace_stride(tot), device="cuda")
        layers, dense, off = [], [], 8
        for o, i in shapes:
            dz, h = torch.randn(n, o, device="cuda", generator=g), torch.randn(n, i, device="cuda", generator=g)
            dense.append((dz, h, off))
            layers.append((engine.quad_pack(dz), engine.quad_pack(h), o, i, off))
            assert torch.equal(engine.quad_unpack(layers[-1][0], n, o), dz)
            off += (o * i + 7) // 4 * 4
        op = engine.DwGemm(layers, flat, ws, ks)
        op()
        first = flat.clone()
        for dz, h, o in dense:
            ref = dz.double().t() @ h.double()
            got = flat[o:o + ref.numel()].view_as(ref).double()
            assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6
        assert float(flat[:8].min()) == 7.0 and float(flat[off:].min()) == 7.0          # nothing outside the layers' ranges is touched
        ws.fill_(float("nan")); op()
        assert torch.equal(flat, first)
    with pytest.raises(engine.OdkError):
        engine.DwGemm([(torch.zeros(32 * 8, device="cuda"), torch.zeros(32 * 8, device="cuda"), 8, 8, 0)], torch.zeros(64, device="cuda"),
                      torch.zeros(8 * 64, device="cuda"), 8)                              # fewer row groups than slices
    # the finishing launch's extra duties: bias gradients from tile sums, partial sums of the squared norm, step count
    n, (o, i) = 640, (40, 52)
    dz, h = torch.randn(n, o, device="cuda", generator=g), torch.randn(n, i, device="cuda", generator=g)
    tiles = [(torch.randn(37, 28, device="cuda", generator=g), 37), (torch.randn(5, 1, device="cuda", generator=g), 5), (torch.randn(40, 512, device="cuda", generator=g), 40)]
    flat = torch.zeros(o * i, device="cuda")
    bias = [torch.full((t.shape[1],), 3.0, device="cuda") for t, _ in tiles]
    acc = torch.zeros(engine.ADAM_ACC_FLOATS, device="cuda"); acc[1] = 4.0
    ws = torch.empty(8 * engine.DwGemm.workspace_stride(o * i), device="cuda")
    op = engine.DwGemm([(engine.quad_pack(dz), engine.quad_pack(h), o, i, 0)], flat, ws, 8, bias=[(t.reshape(-1), b_, nb) for (t, nb), b_ in zip(tiles, bias)], acc=acc)
    op()
    assert float(acc[1]) == 5.0 and 0 < op.norm_blocks <= 1024
    for (t, _), b_ in zip(tiles, bias):
        torch.testing.assert_close(b_, t.sum(0), rtol=1e-5, atol=1e-5)
    want = float(flat.double().square().sum() + sum(b_.double().square().sum() for b_ in bias))
    got = float(acc[2:2 + op.norm_blocks].double().sum())
    assert abs(got - want) < 1e-5 * want and float((flat.view(o, i).double() - dz.double().t() @ h.double()).abs().max()) < 1e-3


def _mlp_params(n_in, n_out, g):
    """Random swish MLP n_in -> 512 -> 256 -> 128 -> n_out as the learner keeps it: one flat buffer (W1 b1 W2 b2 ...), the weight
    table, and the two packed copies built by odk_pack_weights."""
    from open_duck_playground_amd import engine
    widths = (n_in,) + engine.MLP_HIDDEN + (n_out,)
    W = [torch.randn(widths[l + 1], widths[l], device="cuda", generator=g) * (1.5 / widths[l] ** 0.5) for l in range(4)]
    b = [0.3 * torch.randn(widths[l + 1], device="cuda", generator=g) for l in range(4)]
    offs, off = [], 0
    for l in range(4):
        offs.append(off); off += W[l].numel() + b[l].numel()
    flat = torch.cat([t.reshape(-1) for l in range(4) for t in (W[l], b[l])])
    table = engine.WeightTable([(offs[l], widths[l + 1], widths[l], l > 0) for l in range(4)])
    pf, pb = torch.zeros(table.fwd_size, device="cuda"), torch.zeros(table.bwd_size, device="cuda")
    engine.pack_weights(flat, pf, pb, table)
    return widths, W, b, flat, table, pf, pb


def _packed_reference(Wk):
    """[K, N] matrix (reduction index first) -> the packed layout [pad16(K) / 4][N][4], zero padding."""
    K, N = Wk.shape
    K16 = (K + 15) // 16 * 16
    full = torch.zeros(K16, N, device=Wk.device)
    full[:K] = Wk
    return full.view(K16 // 4, 4, N).permute(0, 2, 1).contiguous().reshape(-1)


@pytest.mark.parametrize("n, n_in, n_out", [(320, 101, 28), (336, 212, 1), (5120, 85, 28), (77, 153, 1)])   # whole / ragged tiles, odd and even K, Joystick and Standing sizes
def test_fused_mlp_matches_torch(n, n_in, n_out):
    """odk_mlp_forward / odk_mlp_backward (one launch per direction for the whole swish MLP) vs float64 torch: output, hidden
    activations, swish', every dz, and the bias gradients through odk_colsum_fold; inference-only mode writes `out` alone;
    the packed weight copies against their definition."""
    from open_duck_playground_amd import engine
    g = torch.Generator(device="cuda").manual_seed(n + n_in)
    widths, W, b, flat, table, pf, pb = _mlp_params(n_in, n_out, g)
    for l in range(4):
        assert torch.equal(table.fwd_view(pf, l), _packed_reference(W[l].t()))
        if l > 0:
            assert torch.equal(table.bwd_view(pb, l), _packed_reference(W[l]))
    assert table.bwd_view(pb, 0) is None
    x = torch.randn(n, n_in, device="cuda", generator=g)
    dout = torch.randn(n, n_out, device="cuda", generator=g)
    tiles = (n + 15) // 16
    buf = lambda w: torch.full((n, w), float("nan"), device="cuda")
    wf, wb = [table.fwd_view(pf, l) for l in range(4)], [table.bwd_view(pb, l) for l in range(4)]
    tb = engine.FusedMLP.train_buffers(n, n_in, n_out, "cuda")
    for t in [tb["xp"], tb["doutp"]] + tb["h"] + tb["g"] + tb["dz"] + tb["bias_partial"]:
        t.fill_(float("nan"))
    raw = dict(x=x, wf=wf, wb=wb, b=b, out=buf(n_out), dout=dout, **tb)
    op = engine.FusedMLP([raw])
    op.forward(); op.backward()
    # quad-row buffers: rows past the batch are zeros (the weight-gradient launch reads whole tiles); unpack the rest
    np_ = engine.quad_rows(n)
    for key in ("h", "g", "dz"):
        for l, w in enumerate(engine.MLP_HIDDEN):
            assert float(engine.quad_unpack(raw[key][l], np_, w)[n:].abs().sum()) == 0.0
    assert torch.equal(engine.quad_unpack(raw["xp"], n, n_in), x) and torch.equal(engine.quad_unpack(raw["doutp"], n, n_out), dout)
    assert float(engine.quad_unpack(raw["doutp"], np_, n_out)[n:].abs().sum()) == 0.0
    net = dict(out=raw["out"], bias_partial=raw["bias_partial"], **{key: [engine.quad_unpack(raw[key][l], n, w) for l, w in enumerate(engine.MLP_HIDDEN)]
                                                                    for key in ("h", "g", "dz")})
    # float64 reference
    zs, hs = [], [x.double()]
    for l in range(4):
        z = hs[-1] @ W[l].double().t() + b[l].double()
        zs.append(z)
        if l < 3:
            hs.append(z * torch.sigmoid(z))
    rel = lambda a, ref: float((a.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    assert rel(net["out"], zs[3]) < 2e-6
    dz_ref, dzl = [None] * 3, dout.double()
    for l in (2, 1, 0):
        sg = torch.sigmoid(zs[l])
        gref = sg * (1 + zs[l] * (1 - sg))
        assert rel(net["h"][l], hs[l + 1]) < 2e-6 and rel(net["g"][l], gref) < 2e-6
        dzl = (dzl @ W[l + 1].double()) * gref
        dz_ref[l] = dzl
        assert rel(net["dz"][l], dzl) < 3e-6
    gb = [torch.empty(w, device="cuda") for w in widths[1:]]
    engine.ColsumFold([(net["bias_partial"][l], gb[l]) for l in range(4)], tiles)()
    for l in range(3):
        assert rel(gb[l], dz_ref[l].sum(0)) < 3e-6
    assert rel(gb[3], dout.double().sum(0)) < 3e-6
    # inference only: nothing but `out`
    inf = dict(x=x, wf=wf, b=b, out=buf(n_out))
    engine.FusedMLP([inf]).forward()
    assert torch.equal(inf["out"], net["out"])
    with pytest.raises(engine.OdkError):
        engine.FusedMLP([dict(x=torch.zeros(8, 300, device="cuda"), wf=wf, b=b, out=buf(n_out)[:8])])


def test_fused_mlp_two_networks_one_launch_and_adam_keeps_the_packed_copies():
    """Policy and value side by side in one launch == each alone; odk_adam_clip_packed == odk_adam_clip + a fresh packing."""
    from open_duck_playground_amd import engine
    g = torch.Generator(device="cuda").manual_seed(5)
    nets = []
    for n, n_in, n_out in ((320, 101, 28), (336, 212, 1)):
        widths, W, b, flat, table, pf, pb = _mlp_params(n_in, n_out, g)
        tiles = (n + 15) // 16
        x, dout = torch.randn(n, n_in, device="cuda", generator=g), torch.randn(n, n_out, device="cuda", generator=g)
        mk = lambda: dict(x=x, wf=[table.fwd_view(pf, l) for l in range(4)], wb=[table.bwd_view(pb, l) for l in range(4)], b=b,
                          out=torch.empty(n, n_out, device="cuda"), dout=dout, **engine.FusedMLP.train_buffers(n, n_in, n_out, "cuda"))
        nets.append((mk(), mk()))
    pair = engine.FusedMLP([nets[0][0], nets[1][0]])
    pair.forward(); pair.backward()
    for k in range(2):
        one = engine.FusedMLP([nets[k][1]])
        one.forward(); one.backward()
        for key in ("h", "g", "dz", "bias_partial"):
            assert all(torch.equal(a, b_) for a, b_ in zip(nets[k][0][key], nets[k][1][key]))
        for key in ("out", "xp", "doutp"):
            assert torch.equal(nets[k][0][key], nets[k][1][key])
    # Adam with the packed copies: a 64 x 64 weight (both copies), a bias, a 40 x 21 weight (forward copy only), a bias
    npar = 4096 + 64 + 40 * 21 + 40
    table = engine.WeightTable([(0, 64, 64, True), (4096 + 64, 40, 21, False)])
    p = torch.randn(npar, device="cuda", generator=g); gr = torch.randn(npar, device="cuda", generator=g)
    p2 = p.clone()
    pf, pb = torch.zeros(table.fwd_size, device="cuda"), torch.zeros(table.bwd_size, device="cuda")
    m1, v1, a1 = torch.zeros(npar, device="cuda"), torch.zeros(npar, device="cuda"), torch.zeros(engine.ADAM_ACC_FLOATS, device="cuda")
    m2, v2, a2 = m1.clone(), v1.clone(), a1.clone()
    for _ in range(3):
        engine.adam_clip(p, gr, m1, v1, a1, 3e-4, 1.0)
        engine.adam_clip_packed(p2, gr, m2, v2, a2, pf, pb, table, 3e-4, 1.0)
    # (two compilations of the same arithmetic: fused multiply-adds may differ in the last bit -- every replica runs the same one)
    for got, ref in ((p2, p), (m2, m1), (v2, v1)):
        torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-7)
    assert torch.equal(a1[:2], a2[:2])
    rf, rb = torch.zeros_like(pf), torch.zeros_like(pb)
    engine.pack_weights(p2, rf, rb, table)
    assert torch.equal(pf, rf) and torch.equal(pb, rb)
    assert torch.equal(table.fwd_view(pf, 0), _packed_reference(p2[:4096].view(64, 64).t()))
    assert torch.equal(table.bwd_view(pb, 0), _packed_reference(p2[:4096].view(64, 64)))
    assert torch.equal(table.fwd_view(pf, 1), _packed_reference(p2[4160:4160 + 840].view(40, 21).t()))


@pytest.mark.parametrize("normalize_advantage, N, fused", [(True, 64, True), (False, 64, True), (True, 1024, True), (True, 64, False),
                                                          (True, 64, None)])
def test_flat_learner_gradients_match_autograd(normalize_advantage, N, fused, monkeypatch):
    """loss scalars and every parameter gradient of the captured step == autograd of ppo_loss (same entropy noise): on the
    whole-network kernels (csrc/odk_mlp.hip; N = 1024: the reference minibatch, 256 x 20 rows), on the library path (one GEMM
    per layer: `fused` false), and for an architecture the fused kernels are not built for (`fused` None: falls back)."""
    from open_duck_playground_amd.ppo import learner as LM
    from open_duck_playground_amd.ppo import train as T
    from open_duck_playground_amd.ppo.learner import FlatLearner, prepare_rollout
    from open_duck_playground_amd.ppo.networks import PPONetworks
    dev = torch.device("cuda")
    torch.manual_seed(0)
    if fused is False:
        monkeypatch.setattr(LM, "_FUSED_MLP", False)
    net = (PPONetworks(101, 212, 14) if fused is not None else PPONetworks(101, 212, 14, policy_hidden=(256, 128), value_hidden=(256, 256, 64))).to(dev)
    cfg = T.ppo_config(); cfg["normalize_advantage"] = normalize_advantage
    Tn, nmb = 20, 4
    data = _fake_rollout(N, Tn, dev)
    net.norm_obs.update(data["obs"]); net.norm_priv.update(data["priv"])
    ref = copy.deepcopy(net)
    lr = FlatLearner(net, cfg, N // nmb, Tn, use_graph=False)
    idx = torch.arange(3, 3 + N // nmb, device=dev)
    lr.load_minibatch(prepare_rollout(net, data, cfg), idx)
    assert (lr.fused is not None) == (fused is True)            # the reference architecture runs on the fused network kernels
    lr._draw_noise(); lr._loss_and_grads()
    mb = {k: v[idx] for k, v in data.items()}
    mb["noise"] = lr.noise.view(N // nmb, Tn, 14).clone()
    loss, met = T.ppo_loss(ref, mb, cfg)
    loss.backward()
    got = lr.last_step_losses()
    torch.testing.assert_close(got[0], loss.detach(), rtol=2e-4, atol=1e-6)
    torch.testing.assert_close(got[1], met["policy_loss"], rtol=2e-4, atol=1e-6)
    torch.testing.assert_close(got[2], met["v_loss"], rtol=2e-4, atol=1e-6)
    torch.testing.assert_close(got[3], met["entropy_loss"], rtol=2e-4, atol=1e-6)
    ref_params = list(ref.policy.parameters()) + list(ref.value.parameters())
    flat_ref = torch.cat([p.grad.reshape(-1) for p in ref_params])
    err = (lr.flat_g - flat_ref).abs().max() / flat_ref.abs().max()
    assert err < 2e-4, float(err)


def test_indexed_step_equals_the_gathered_step(monkeypatch):
    """Round 5: the minibatch step WITHOUT a gathered copy -- forward rows through the schedule's trajectory indices
    (odk_mlp_desc.row_idx), GAE + statistics + loss head in one launch (odk_ppo_gae_head), the cursor advanced by the clip + Adam
    launch -- against the gathered form (odk_gather_rows + odk_gae + odk_ppo_head): same arithmetic, same orders, so gradients,
    advantages and parameters agree BIT FOR BIT, step after step along a two-pass schedule; an index outside the rollout poisons its
    trajectory with NaN instead of reading out of bounds."""
    from open_duck_playground_amd.ppo import learner as LM
    from open_duck_playground_amd.ppo import train as T
    from open_duck_playground_amd.ppo.learner import FlatLearner, prepare_rollout
    from open_duck_playground_amd.ppo.networks import PPONetworks
    dev = torch.device("cuda")
    cfg = T.ppo_config(); cfg.update(num_minibatches=4, num_updates_per_batch=2, tune_gemms=False)
    N, Tn = 64, 20
    data = _fake_rollout(N, Tn, dev, seed=9)
    nets, lrs = [], []
    for indexed in ("1", "0"):
        monkeypatch.setenv("ODK_LEARNER_INDEXED", indexed)
        torch.manual_seed(4)
        n = PPONetworks(101, 212, 14).to(dev)
        n.norm_obs.update(data["obs"]); n.norm_priv.update(data["priv"])
        nets.append(n)
        lrs.append(FlatLearner(n, cfg, N // 4, Tn, use_graph=True))
    a, b = lrs
    assert a.indexed and not b.indexed and a.gae_head is not None
    g = torch.Generator(device=dev).manual_seed(0)
    perms = torch.cat([torch.randperm(N, generator=g, device=dev) for _ in range(2)])
    # indexed: one schedule, eight replays; gathered: eight load_minibatch + step, fed the SAME noise
    a.load_rollout_from(nets[0], data, cfg)
    a.set_schedule(perms)
    prep = prepare_rollout(nets[1], data, cfg)
    for k, v in prep.items():
        assert torch.equal(a.roll[k][:N], v), k                # prepare_rollout(out=...) == prepare_rollout
    for k in range(8):
        b.load_minibatch(prep, perms[k * 16:(k + 1) * 16].contiguous())
        b.noise.copy_(a.noise)                                  # a.noise = the pool slot under a's cursor
        a.step(); b.step()
        assert torch.equal(a.flat_g, b.flat_g), k
        assert torch.equal(a.adv, b.adv) and torch.equal(a.vs, b.vs) and torch.equal(a.stats, b.stats), k
        assert torch.equal(a.flat_p, b.flat_p), k
    assert int(a.cursor) == 8 and float(a.acc[1]) == 8.0
    torch.testing.assert_close(a.losses, b.losses, rtol=1e-5, atol=1e-6)      # (sums of float atomics: order differs)
    with pytest.raises(LM.engine.OdkError, match="used up"):
        a.step()
    # the compatibility path of the indexed learner: load_minibatch == a one-step schedule
    idx = perms[:16].contiguous()
    a.load_minibatch(prep, idx); b.load_minibatch(prep, idx)
    b.noise.copy_(a.noise)
    a.step(); b.step()
    assert torch.equal(a.flat_p, b.flat_p) and int(a.cursor) == 1
    # an index outside the resident rollout is never dereferenced
    bad = idx.clone(); bad[3] = a.cap + 5
    a.load_minibatch(prep, bad)
    a._loss_and_grads()
    assert not torch.isfinite(a.last_step_losses()).all()
    # K steps as one graph: `run` == the same steps one by one (two fresh learners, K = 4 through the environment)
    monkeypatch.setenv("ODK_LEARNER_INDEXED", "1"); monkeypatch.setenv("ODK_LEARNER_STEPS_PER_GRAPH", "4")
    pair = []
    for _ in range(2):
        torch.manual_seed(4)
        n = PPONetworks(101, 212, 14).to(dev)
        n.norm_obs.update(data["obs"]); n.norm_priv.update(data["priv"])
        pair.append((n, FlatLearner(n, cfg, N // 4, Tn, use_graph=True)))
    (n1, l1), (n2, l2) = pair
    assert l1.graph_k is not None and l1.K == 4
    for (nn, ll) in pair:
        ll.load_rollout_from(nn, data, cfg); ll.set_schedule(perms)
    l2._pool.copy_(l1._pool)
    l1.run(7)                                  # one 4-step replay + three single steps
    for _ in range(7):
        l2.step()
    assert torch.equal(l1.flat_p, l2.flat_p) and int(l1.cursor) == int(l2.cursor) == 7 and l1.nsteps == l2.nsteps == 7
    torch.testing.assert_close(l1.losses, l2.losses, rtol=0, atol=0)


def test_col_moments_kernel_matches_float64_torch():
    """odk_col_moments (the observation normaliser's batch statistics in one pass, float64 accumulators) vs torch in float64; and
    RunningStats.update through it == the torch path it replaces, to rounding."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.ppo.networks import RunningStats
    g = torch.Generator(device="cuda").manual_seed(0)
    for rows, w in ((163840, 101), (5000, 212), (4097, 7), (33, 70)):
        x = (torch.randn(rows, w, device="cuda", generator=g) * 3 + 1.5).contiguous()
        s, s2 = engine.col_moments(x)
        xd = x.double()
        torch.testing.assert_close(s, xd.sum(0), rtol=1e-12, atol=1e-9)
        torch.testing.assert_close(s2, (xd * xd).sum(0), rtol=1e-12, atol=1e-9)
        a, b = engine.col_moments(x)
        assert torch.equal(a, s) and torch.equal(b, s2)                    # fixed order: bit-reproducible
    x = torch.randn(8192, 20, 101, device="cuda", generator=g) * 2 + 0.5
    fast, slow = RunningStats(101).cuda(), RunningStats(101).cuda()
    for k in range(2):
        fast.update(x + k)
        xs = (x + k).reshape(-1, 101)
        n = torch.tensor([float(xs.shape[0])], dtype=torch.float64, device="cuda")
        # the torch arithmetic of RunningStats.update on exact float64 sums
        s, s2 = xs.double().sum(0), (xs.double() ** 2).sum(0)
        count = slow.count + n[0]; delta = s / n[0] - slow.mean.double(); new_mean = slow.mean.double() + delta * (n[0] / count)
        sv = slow.summed_variance.double() + (s2 - s * (slow.mean.double() + new_mean) + n[0] * slow.mean.double() * new_mean)
        slow.count.copy_(count); slow.mean.copy_(new_mean.float()); slow.summed_variance.copy_(sv.float())
        slow.std.copy_(torch.sqrt(torch.clamp(sv / count, min=0)).float().clamp(1e-6, 1e6))
    torch.testing.assert_close(fast.mean, slow.mean, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(fast.std, slow.std, rtol=1e-6, atol=1e-7)


def test_indexed_learner_grows_its_resident_rollout():
    """A rollout larger than the resident copy was sized for (n_traj): new buffers, new descriptors, new graphs -- and the same
    parameters afterwards as a learner that was built large enough."""
    from open_duck_playground_amd.ppo import train as T
    from open_duck_playground_amd.ppo.learner import FlatLearner
    from open_duck_playground_amd.ppo.networks import PPONetworks
    dev = torch.device("cuda")
    cfg = T.ppo_config(); cfg.update(num_minibatches=4, num_updates_per_batch=1, tune_gemms=False)
    N, Tn = 64, 20
    data = _fake_rollout(N, Tn, dev, seed=12)
    out = []
    for n_traj in (16, 64):
        torch.manual_seed(6)
        n = PPONetworks(101, 212, 14).to(dev)
        n.norm_obs.update(data["obs"]); n.norm_priv.update(data["priv"])
        lr = FlatLearner(n, cfg, N // 4, Tn, use_graph=True, n_traj=n_traj)
        assert lr.indexed and lr.cap == n_traj
        lr.load_rollout_from(n, data, cfg)
        assert lr.cap == 64 and lr.graph_a is not None
        g = torch.Generator(device=dev).manual_seed(1)
        lr.set_schedule(torch.randperm(N, generator=g, device=dev))
        out.append(lr)
    out[1]._pool.copy_(out[0]._pool)
    for lr in out:
        lr.run(4)
    assert torch.equal(out[0].flat_p, out[1].flat_p) and int(out[0].cursor) == 4


def test_flat_learner_training_step_matches_eager_and_graph_replays():
    """3 clipped-Adam steps: graph replay == plain launches == autograd + optax-style clip + torch Adam."""
    from open_duck_playground_amd.ppo import train as T
    from open_duck_playground_amd.ppo.learner import FlatLearner, prepare_rollout
    from open_duck_playground_amd.ppo.networks import PPONetworks
    dev = torch.device("cuda")
    cfg = T.ppo_config(); cfg["max_grad_norm"] = 0.05   # make the clip bite
    cfg["tune_gemms"] = False                           # keep the three learners on the same GEMM kernels (and the test short)
    N, Tn, nmb = 64, 20, 4
    data = _fake_rollout(N, Tn, dev, seed=3)
    nets = []
    for _ in range(3):
        torch.manual_seed(1)
        n = PPONetworks(101, 212, 14).to(dev)
        n.norm_obs.update(data["obs"]); n.norm_priv.update(data["priv"])
        nets.append(n)
    ref = nets[2]
    before = torch.cat([p.detach().reshape(-1).clone() for p in list(ref.policy.parameters()) + list(ref.value.parameters())])
    learners = [FlatLearner(nets[0], cfg, N // nmb, Tn, use_graph=True), FlatLearner(nets[1], cfg, N // nmb, Tn, use_graph=False)]
    torch.testing.assert_close(learners[0].flat_p, before)     # graph warm-up must not train
    opt = torch.optim.Adam(list(ref.policy.parameters()) + list(ref.value.parameters()), lr=cfg["learning_rate"])
    prep = prepare_rollout(ref, data, cfg)
    for k in range(3):
        idx = torch.arange(k * 16, k * 16 + 16, device=dev)
        noise = torch.randn(16 * Tn, 14, device=dev)
        learners[1].load_minibatch(prep, idx)
        learners[1].noise.copy_(noise)
        learners[1].sample_noise = False    # use the injected entropy sample
        learners[1].step()
        mb = {kk: v[idx] for kk, v in data.items()}
        mb["noise"] = noise.view(16, Tn, 14)
        loss, _ = T.ppo_loss(ref, mb, cfg)
        opt.zero_grad(); loss.backward()
        T.clip_by_global_norm(list(ref.policy.parameters()) + list(ref.value.parameters()), cfg["max_grad_norm"])
        opt.step()
    after_ref = torch.cat([p.detach().reshape(-1) for p in list(ref.policy.parameters()) + list(ref.value.parameters())])
    moved = (after_ref - before).abs().max()
    assert moved > 1e-4
    assert (learners[1].flat_p - after_ref).abs().max() < 2e-3 * moved + 1e-7
    # graph replay: same code path with in-graph noise; parameters move by a comparable amount and stay finite
    for k in range(3):
        learners[0].load_minibatch(prep, torch.arange(k * 16, k * 16 + 16, device=dev))
        learners[0].step()
    torch.cuda.synchronize()
    d0 = (learners[0].flat_p - before).abs().max()
    assert torch.isfinite(learners[0].flat_p).all() and 0.2 * moved < d0 < 5 * moved
    assert float(learners[0].acc[1]) == 3.0
    # the module parameters ARE the flat buffer (rollout policy sees the update)
    assert nets[0].policy.layers[0].weight.data_ptr() == learners[0].flat_p.data_ptr()


def _dp_worker(rank, world, port, out):
    """Two ranks on one GPU over gloo (RCCL refuses two ranks per device): the data-parallel FlatLearner path --
    graph (loss + grads), all-reduce of the flat gradient, graph (clip + Adam)."""
    import os
    import torch.distributed as dist
    from open_duck_playground_amd.ppo import train as T
    from open_duck_playground_amd.ppo.learner import FlatLearner, prepare_rollout
    from open_duck_playground_amd.ppo.networks import PPONetworks
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = PPONetworks(101, 212, 14).to(dev)
    data = _fake_rollout(32, 20, dev, seed=10 + rank)          # each rank has its own shard
    net.norm_obs.update(data["obs"], dist.group.WORLD); net.norm_priv.update(data["priv"], dist.group.WORLD)
    cfg = T.ppo_config(); cfg.update(num_minibatches=2, num_updates_per_batch=2, tune_gemms=False)
    before = torch.cat([p.detach().reshape(-1).clone() for p in list(net.policy.parameters()) + list(net.value.parameters())])
    lr = FlatLearner(net, cfg, 16, 20, world=world, group=dist.group.WORLD)
    assert lr.graph_b is not None
    m = T.sgd_epoch(net, None, data, cfg, torch.Generator(device=dev).manual_seed(3), world=world, learner=lr)
    torch.cuda.synchronize()
    flat = lr.flat_p.detach().cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out.put((bool(torch.equal(gathered[0], gathered[1])), float((flat - before.cpu()).abs().max()), bool(torch.isfinite(flat).all()),
                 float(lr.acc[1]), float(m["total_loss"])))
    dist.destroy_process_group()


def test_flat_learner_data_parallel_two_ranks():
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    same, moved, finite, steps, loss = q.get(timeout=300)
    for p in procs: p.join(timeout=60)
    assert same, "ranks diverged: the flat gradient all-reduce must make the updates identical"
    assert finite and moved > 1e-5 and steps == 4.0 and loss == loss


def test_policy_sample_kernel_matches_the_torch_distribution():
    """raw action, tanh action and log-density of the fused rollout sampler == the torch formulas the CPU path uses."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.ppo.networks import tanh_normal_log_prob
    import torch.nn.functional as F
    g = torch.Generator(device="cuda").manual_seed(5)
    n, A = 1000, 14                                        # n not a multiple of the 16 samples per workgroup
    logits = 2.0 * torch.randn(n, 2 * A, device="cuda", generator=g)
    for z in (torch.randn(n, A, device="cuda", generator=g), torch.zeros(n, A, device="cuda")):
        raw, act, logp = engine.policy_sample(logits, z)
        loc, scale = logits[:, :A], F.softplus(logits[:, A:]) + 0.001
        raw_ref = loc + scale * z
        torch.testing.assert_close(raw, raw_ref, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(act, torch.tanh(raw_ref), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(logp, tanh_normal_log_prob(loc, scale, raw_ref), rtol=1e-4, atol=2e-3)


def test_learner_metrics_are_means_over_all_steps_since_the_last_call():
    """brax reports the mean loss over all SGD steps of an epoch: the loss head adds into running sums, metrics() divides
    by the step count and resets."""
    from open_duck_playground_amd.ppo import train as T
    from open_duck_playground_amd.ppo.learner import FlatLearner, prepare_rollout
    from open_duck_playground_amd.ppo.networks import PPONetworks
    dev = torch.device("cuda")
    torch.manual_seed(2)
    net = PPONetworks(101, 212, 14).to(dev)
    cfg = T.ppo_config(); cfg["tune_gemms"] = False
    data = _fake_rollout(64, 20, dev, seed=4)
    net.norm_obs.update(data["obs"]); net.norm_priv.update(data["priv"])
    lr = FlatLearner(net, cfg, 16, 20, use_graph=True)
    assert float(lr.losses.abs().sum()) == 0.0                 # the capture warm-up steps leave no loss behind
    prep = prepare_rollout(net, data, cfg)
    per_step = []
    for k in range(4):
        lr.load_minibatch(prep, torch.arange(k * 16, k * 16 + 16, device=dev))
        before = lr.losses.clone()
        lr.step()
        per_step.append(lr.losses - before)
    m = lr.metrics()
    ref = torch.stack(per_step).mean(0)
    got = torch.stack([m["total_loss"], m["policy_loss"], m["v_loss"], m["entropy_loss"]])
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-6)
    assert lr.nsteps == 0 and float(lr.losses.abs().sum()) == 0.0
    assert torch.stack(per_step)[:, 2].std() > 0               # the steps differ, so a mean is not the last step's value


def test_split_update_over_rccl_matches_the_single_graph_step():
    """The data-parallel step (graph A: loss + gradients -> RCCL all-reduce of the flat gradient on the process group's
    stream -> graph B: clip + Adam) on the real `nccl` backend.  One GPU admits one rank, so the group has world size 1
    (the all-reduce is the identity): what is pinned is the stream hand-over graph -> collective -> graph without host
    synchronisation -- parameters after 6 steps are bit-identical to the single-graph learner fed the same noise."""
    import os
    import torch.distributed as dist
    from open_duck_playground_amd.ppo import train as T
    from open_duck_playground_amd.ppo.learner import FlatLearner, prepare_rollout
    from open_duck_playground_amd.ppo.networks import PPONetworks
    dev = torch.device("cuda", 0)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(32500 + os.getpid() % 2000)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        cfg = T.ppo_config(); cfg["tune_gemms"] = False
        data = _fake_rollout(64, 20, dev, seed=6)
        nets = []
        for _ in range(2):
            torch.manual_seed(3)
            n = PPONetworks(101, 212, 14).to(dev)
            n.norm_obs.update(data["obs"], dist.group.WORLD); n.norm_priv.update(data["priv"], dist.group.WORLD)
            nets.append(n)
        one = FlatLearner(nets[0], cfg, 16, 20, fused_norm=False)     # (the norm summed as the split path sums it: bit-comparable)
        two = FlatLearner(nets[1], cfg, 16, 20, world=1, group=dist.group.WORLD, split_update=True)
        assert one.graph_b is None and two.graph_b is not None
        prep = prepare_rollout(nets[0], data, cfg)
        for k in range(6):
            idx = torch.arange((k % 4) * 16, (k % 4) * 16 + 16, device=dev)
            one.load_minibatch(prep, idx)
            two.load_minibatch(prep, idx)
            two.noise.copy_(one.noise)                                   # same entropy noise
            one.step()
            two.step()
        T.assert_replicas_identical(nets[1], dist.group.WORLD)
        torch.cuda.synchronize()
        assert torch.equal(one.flat_p, two.flat_p) and torch.equal(one.m, two.m)
        assert float(two.acc[1]) == 6.0 and torch.isfinite(two.flat_p).all()
        m1, m2 = one.metrics(), two.metrics()
        assert abs(float(m1["v_loss"]) - float(m2["v_loss"])) < 1e-6
    finally:
        dist.destroy_process_group()


def test_row_gather_refuses_bad_indices():
    from open_duck_playground_amd import engine
    src = torch.arange(40, dtype=torch.float32, device="cuda").view(10, 4)
    dst = torch.zeros(3, 4, device="cuda")
    g = engine.RowGather([(src, dst)])
    g(torch.tensor([7, 0, 9], device="cuda"))
    assert dst.tolist() == [src[7].tolist(), src[0].tolist(), src[9].tolist()]
    for bad in (torch.tensor([7, 0, 9], device="cuda", dtype=torch.int32), torch.tensor([7, 0, 9]), torch.tensor([1, 2], device="cuda"),
                torch.arange(6, device="cuda")[::2]):
        with pytest.raises(engine.OdkError):
            g(bad)
    g(torch.tensor([1, 10, -1], device="cuda"))                # out of range: NaN rows, never an out-of-bounds read
    assert dst[0].tolist() == src[1].tolist() and torch.isnan(dst[1:]).all()
    # rows longer than one 1024-float chunk, odd row lengths (no 16-byte path), and a direct (un-indexed) block field
    big, odd, pool = torch.randn(9, 2500, device="cuda"), torch.randn(9, 7, device="cuda"), torch.randn(12, 40, device="cuda")
    dbig, dodd, dpool = torch.zeros(3, 2500, device="cuda"), torch.zeros(3, 7, device="cuda"), torch.zeros(3, 40, device="cuda")
    g2 = engine.RowGather([(big, dbig), (odd, dodd)], [(pool, dpool)])
    pick = torch.tensor([8, 2, 5], device="cuda")
    g2(pick, [6])
    assert torch.equal(dbig, big[pick]) and torch.equal(dodd, odd[pick]) and torch.equal(dpool, pool[6:9])
    with pytest.raises(engine.OdkError):
        g2(pick, [10])                                         # direct block past the end of its source
    with pytest.raises(engine.OdkError):
        engine.RowGather([(src, dst), (torch.zeros(9, 4, device="cuda"), torch.zeros(3, 4, device="cuda"))])


def test_gae_treats_any_nonzero_flag_as_set_in_every_kernel():
    from open_duck_playground_amd import engine
    g = torch.Generator(device="cuda").manual_seed(1)
    for B, Tn in ((256, 20), (300, 20), (1200, 40)):           # LDS-staged, register-staged, generic
        rew, val = torch.randn(B, Tn, device="cuda", generator=g), torch.randn(B, Tn, device="cuda", generator=g)
        boot = torch.randn(B, device="cuda", generator=g)
        term = (torch.rand(B, Tn, device="cuda", generator=g) < 0.1).float()
        trunc = (torch.rand(B, Tn, device="cuda", generator=g) < 0.05).float() * (1 - term)
        a = engine.gae(trunc, term, rew, val, boot, 0.95, 0.97)
        b = engine.gae(trunc * 3.0, term * 0.25, rew, val, boot, 0.95, 0.97)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_fused_policy_inference_matches_the_module():
    """Rollout-side policy inference through the whole-network kernel == the torch module (same parameters, after an update too)."""
    from open_duck_playground_amd.ppo.learner import fused_policy
    from open_duck_playground_amd.ppo.networks import PPONetworks
    torch.manual_seed(2)
    net = PPONetworks(101, 212, 14).cuda()
    obs = torch.randn(300, 101, device="cuda")
    net.norm_obs.update(obs * 3 + 1)
    fp = fused_policy(net, 300)
    assert fp is not None and fused_policy(net, 300) is fp
    for rnd in range(2):
        fp.refresh()
        ref = net.policy(net.norm_obs(obs))
        torch.testing.assert_close(fp(obs), ref, rtol=2e-5, atol=2e-6)      # raw observations in: the normaliser runs inside the kernel's load
        with torch.no_grad():
            for p_ in net.policy.parameters():
                p_.mul_(1.01)
    assert fused_policy(PPONetworks(101, 212, 14, policy_hidden=(64, 64)).cuda(), 8) is None


def test_tiny_minibatch_falls_back_to_the_library_path():
    """A minibatch of fewer than 128 samples (8 rows per slice of the weight-gradient launch) runs the library path."""
    from open_duck_playground_amd.ppo import train as T
    from open_duck_playground_amd.ppo.learner import FlatLearner, prepare_rollout
    from open_duck_playground_amd.ppo.networks import PPONetworks
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = PPONetworks(101, 212, 14).to(dev)
    cfg = T.ppo_config(); cfg["tune_gemms"] = False
    data = _fake_rollout(8, 5, dev)
    lr = FlatLearner(net, cfg, 4, 5, use_graph=False)
    assert lr.fused is None
    lr.load_minibatch(prepare_rollout(net, data, cfg), torch.arange(4, device=dev))
    lr.step()
    assert torch.isfinite(lr.flat_p).all() and float(lr.acc[1]) == 1.0
