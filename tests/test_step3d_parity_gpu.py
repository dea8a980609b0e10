"""Whole 3-D training step on the HIP path (arco_amd.train_arco_3d.ArcoStep3D.step) vs the CPU oracle of
train_arco_3d.py:255-400 (oracle/cpu_step3d.py): same weights, same CPU-generator seed, Dropout3d off, three chained steps -
iteration 0 optimises `unsup + supervised + loss_eqv` (:393), iterations 1.. `k1 reco + k3 unsup + supervised` (:391).
The oracle step takes the step's gradient-free decision inputs (pseudo-labels, entropy masks, teacher probabilities) from the HIP
step after checking them against its own (cpu_step3d.step `force`): with 10^5 voxels a few always sit within fp32 rounding of a
threshold, and one flipped voxel changes a sampler argument and with it every later draw of the CPU generator.
Checked: those inputs; every loss term, banks / pointers, the updated student, heads and teacher, and the BatchNorm running statistics of
both V-Nets, which must receive their momentum updates in the reference's pass order (-m gpu)."""
import random

import numpy as np
import pytest
import torch

import cpu_step3d
import fixture_inputs as fx

pytestmark = pytest.mark.gpu

FEA = (128, 64, 32, 16, 16)


def _drop_off(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout3d):
            mod.p = 0.0
    m.has_dropout = False


def _qrep_w(seed):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal((16, 16, 1, 1, 1)) / 4.0).astype(np.float32))


def _volumes(rs, b, patch, C):
    l = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
    u = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
    lab = torch.from_numpy(fx.blob_labels(rs, b, patch, C))
    l = l + 0.5 * (lab > 0).unsqueeze(1).float()           # some structure for the nets to follow
    return l, lab, u


def _state(C):
    """fx.vnet_state with the logits layer scaled up: a random-init V-Net puts every voxel's class probabilities within a few
    percent of 1 / C, and with 10^5 voxels some pseudo-label arg-max / class-rank decision then sits inside fp32 rounding of a
    tie (seen: one voxel of 65 536 changing its pseudo-label between the CPU and the GPU forward shifts every sampled anchor
    index).  Spread logits keep the comparison well-posed without touching what is compared."""
    sd = fx.vnet_state(52, 1, C)
    sd["out_conv.weight"] = sd["out_conv.weight"] * 4.0
    sd["out_conv.bias"] = sd["out_conv.bias"] * 2.0
    return sd


_STATS = {"e2": 0.0, "emax": 0.0}          # worst update deviations seen (printed by the test: how far from the bounds)


def _check_state(st_g, st_o, it, before):
    """Updated weights.  Loss terms are held to north_star's 1e-3 by the caller; parameters are compared through their
    UPDATES (new - old, old = the common state both sides started the step from): the V-Net's gradient carries the
    ReLU-decision noise every fp32 implementation has (the fp32 reference sits 0.3-2 % from its own float64 run:
    tests/golden g18 `ref32_dev`), so an update is held to 10 % in L2 and 30 % of its largest element (2 volumes per BatchNorm group, 16 values per channel
    at the bottleneck: measured up to 13 % element-wise in the 4^3 / 8^3 levels) - a sign / scale / missing-term check - and the weights themselves to 3e-3; the strict
    gradient checks of the V-Net live in tests/test_nets3d_gpu.py (in-network tensors)."""
    def one(name, got, ref, old, wtol=3e-3):
        got, ref, old = got.detach().cpu(), ref.detach().cpu(), old.detach().cpu()
        scale = max(1e-6, float(ref.abs().max()))
        assert float((got - ref).abs().max()) / scale < wtol, (it, name, float((got - ref).abs().max()) / scale)
        upd = float((ref - old).abs().max())
        if upd > 1e-5 * scale:
            d = (got - old) - (ref - old)
            e2 = float(d.norm()) / float((ref - old).norm())
            _STATS["e2"] = max(_STATS["e2"], e2); _STATS["emax"] = max(_STATS["emax"], float(d.abs().max()) / upd)
            assert e2 <= 0.1, (it, name, "update, L2", e2)
            assert float(d.abs().max()) <= 0.3 * upd + 1e-6 * scale, (it, name, "update, max", float(d.abs().max()) / upd)

    sd_g = st_g.model.state_dict()
    for k, p in st_o["student"].items():
        if p.requires_grad and not (k.endswith(".bias") and ".conv." in k and int(k.split(".")[-2]) % 3 == 0 and "out_conv" not in k):
            one(k, sd_g[k], p, before["student"][k])
    for k, p in st_o["q_fe"].items():
        one("q_fe." + k, st_g.q_feature_extractor.state_dict()[k], p, before["q_fe"][k])
    for i in range(2):
        one(f"q_rep.{i}", st_g.q_representation[i].weight, st_o["q_rep"][i], before["q_rep"][i])
    for k, p in st_o["k_fe"].items():
        one("k_fe." + k, st_g.k_feature_extractor.state_dict()[k], p, before["k_fe"][k], wtol=1e-3)
    sd_t = st_g.ema_model.state_dict()
    for k, p in st_o["teacher"].items():
        if p.is_floating_point() and "running" not in k:
            assert float((sd_t[k].cpu() - p).abs().max()) / max(1e-6, float(p.abs().max())) < 1e-3, (it, k)
    # BatchNorm running statistics in the reference's order (student: l, u_aug [, warped]; teacher: u, l, u_aug)
    for name, sd_ref, sd_got in (("student", st_o["student"], sd_g), ("teacher", st_o["teacher"], sd_t)):
        for k, p in sd_ref.items():
            if "running" in k:
                np.testing.assert_allclose(sd_got[k].cpu().numpy(), p.detach().numpy(), rtol=3e-4, atol=1e-4 * float(p.abs().max()),
                                           err_msg=f"step {it} {name} {k}")


def _snapshot(st_o):
    return dict(student={k: v.detach().clone() for k, v in st_o["student"].items()},
                q_fe={k: v.detach().clone() for k, v in st_o["q_fe"].items()},
                k_fe={k: v.detach().clone() for k, v in st_o["k_fe"].items()},
                q_rep=[w.detach().clone() for w in st_o["q_rep"]])


def _sync_from_oracle(st_g, st_o, bank_o, ptr_o):
    """GPU trainer <- oracle state (weights, BatchNorm buffers, teacher, heads, momentum, banks): every step is compared from
    EQUAL state.  Chained fp32 trajectories of two implementations drift apart - the V-Net's gradients carry ~1e-2 of fp32
    conditioning noise in the reference itself (oracle/gen_golden.py g18) - and a drifted step would test the drift."""
    from arco_amd import ops
    with torch.no_grad():
        st_g.model.load_state_dict({k: v.detach() for k, v in st_o["student"].items()}, strict=True)
        st_g.ema_model.load_state_dict({k: v.detach() for k, v in st_o["teacher"].items()}, strict=True)
        st_g.q_feature_extractor.load_state_dict({k: v.detach() for k, v in st_o["q_fe"].items()}, strict=True)
        st_g.k_feature_extractor.load_state_dict({k: v.detach() for k, v in st_o["k_fe"].items()}, strict=True)
        for i in range(2):
            st_g.q_representation[i].weight.copy_(st_o["q_rep"][i].detach())
        opt = st_g.optimizer
        n_leaves = len(opt.params)
        for i in range(n_leaves):
            off, k = opt.offsets[i]
            if i in st_o["mom"]:
                opt.flat_buf[off:off + k].copy_(st_o["mom"][i].reshape(-1))
                opt._started[i] = True
            else:
                opt._started[i] = False
        for c in range(len(bank_o)):
            st_g.memobank[c] = [bank_o[c][0].clone().cuda()]
            st_g.queue_ptrlis[c] = ptr_o[c].clone()
    ops.bump_weight_epoch()


VARIANTS = [
    dict(tag="dense_c2", n_cls=2, dense_head=1, eqv_pass=1, apply_aug="cutmix", func="asmc", graphs=0),
    dict(tag="sparse_c4", n_cls=4, dense_head=0, eqv_pass=1, apply_aug="cutmix", func="asmc", graphs=0, strong_threshold=0.3),
    dict(tag="sparse_c4_noeqv_smc", n_cls=4, dense_head=0, eqv_pass=0, apply_aug="cutout", func="smc", graphs=0),
    dict(tag="sparse_c4_graphs", n_cls=4, dense_head=0, eqv_pass=1, apply_aug="classmix", func="asmc", graphs=1, strong_threshold=0.3),
    dict(tag="dense_c4_revisit", n_cls=4, revisit=1, K=4, topk=2, eqv_pass=1, apply_aug="cutmix", func="asmc", graphs=0),
]


@pytest.mark.parametrize("variant", VARIANTS, ids=[v["tag"] for v in VARIANTS])
def test_three_steps_3d_vs_cpu_oracle(variant):
    from arco_amd import ops, train_arco_3d as T3
    v = dict(variant)
    v.pop("tag")
    C = v.pop("n_cls")
    b, patch, Q, Nn, qs, lr = 2, (32, 32, 32), 48, 16, 200, 0.01
    vnet_sd = _state(C)
    fe_sd = fx.fe_state(61, FEA, 16, nd=3)
    qrep_w = [_qrep_w(71), _qrep_w(72)]
    argv = ["--batch_size", str(b), "--queue_size", str(qs), "--synthetic", "1", "--num_classes", str(C), "--num_queries", str(Q),
            "--num_negatives", str(Nn), "--k1", "1.0", "--base_lr", str(lr)]
    for k, val in v.items():
        argv += [f"--{k}", str(val)]
    args = T3.build_parser().parse_args(argv)
    args.patch_size = list(patch)
    random.seed(3); np.random.seed(3); torch.manual_seed(3)
    st_g = T3.ArcoStep3D(args, "cuda:0")
    st_g.keep_debug = True
    st_g.model.load_state_dict(vnet_sd, strict=True)
    st_g.ema_model.load_state_dict(vnet_sd, strict=True)
    st_g.q_feature_extractor.load_state_dict(fe_sd, strict=True)
    st_g.k_feature_extractor.load_state_dict(fe_sd, strict=True)
    with torch.no_grad():
        st_g.q_representation[0].weight.copy_(qrep_w[0])
        st_g.q_representation[1].weight.copy_(qrep_w[1])
    for m in (st_g.model, st_g.ema_model):
        _drop_off(m)
    ops.bump_weight_epoch()
    st_o = cpu_step3d.make_state(vnet_sd, fe_sd, qrep_w, base_lr=lr, max_iterations=args.max_iterations)
    # the banks start from the trainer's own randn rows (train_arco_3d.py:148): the oracle gets copies
    bank_o = [[m[0].detach().cpu().clone()] for m in st_g.memobank]
    ptr_o = [torch.zeros(1, dtype=torch.long) for _ in range(C)]
    qsz = list(st_g.queue_size)
    pool_o = None
    if v.get("revisit"):
        assert st_g.random_pool is not None and args.dense_head == 1
        pool_o = dict(rows=st_g.random_pool.channels_first().cpu().clone(), ptr=torch.zeros(1, dtype=torch.long))
    rs = np.random.RandomState(13)
    keys_seen = 0
    for it in range(3):
        l, lab, u = _volumes(rs, b, patch, C)
        before = _snapshot(st_o)
        random.seed(10 + it); np.random.seed(10 + it); torch.manual_seed(10 + it)
        st_g.step(l.cuda(), lab.cuda(), u.cuda())
        force = {k: t.detach().cpu() for k, t in st_g.decisions.items()}
        random.seed(10 + it); np.random.seed(10 + it); torch.manual_seed(10 + it)
        cpu_step3d.step(st_o, l, lab, u, bank_o, ptr_o, qsz, n_cls=C, k1=1.0, k3=args.k3, k4=args.k4, delta_n=args.strong_threshold_u2pl,
                        strong_threshold=args.strong_threshold, weak_threshold=args.weak_threshold, func=v["func"], nq=Q, nn_=Nn,
                        tps_sigma=args.tps_sigma, apply_aug=v["apply_aug"], eqv_pass=bool(v["eqv_pass"]), pool=pool_o,
                        topk=v.get("topk", 5), force=force)
        # the decision inputs themselves: probabilities to 1e-3 (north_star; the x16 logits layer of _state amplifies the forward's 2e-5), the discrete ones equal except at
        # voxels that sit on a threshold / tie within that rounding (measured: < 0.1 % of them)
        ag = st_o["agree"]
        for k in ("pseudo_logits", "prob_l_t", "prob_u_t"):
            assert ag[k]["max_abs_diff"] < 1e-3, (it, k, ag[k])
        for k in ("pseudo_labels", "low", "high"):
            assert ag[k]["n_diff"] <= max(2, 2e-3 * ag[k]["n"]), (it, k, ag[k])
        to, tg = st_o["last_terms"], st_g.last_terms
        names = ("ce", "dice", "unsup", "reco") + (("eqv",) if v["eqv_pass"] else ()) + (("loss_q",) if pool_o is not None else ())
        for k in names:
            np.testing.assert_allclose(float(tg[k]), to[k], rtol=1e-3, atol=1e-5, err_msg=f"step {it} {k}")      # north_star: 1e-3
        assert [int(p) for p in ptr_o] == [int(p) for p in st_g.queue_ptrlis], it
        for bo, bg in zip(bank_o, st_g.memobank):
            assert bo[0].shape == bg[0].shape, it
            np.testing.assert_allclose(bg[0].cpu().numpy(), bo[0].numpy(), rtol=1e-3, atol=2e-4)
        keys_seen += sum(int(x[0].shape[0]) - 1 for x in bank_o)
        if pool_o is not None:
            np.testing.assert_allclose(st_g.random_pool.channels_first().cpu().numpy(), pool_o["rows"].numpy(), rtol=2e-3, atol=2e-6)
            assert int(st_g.random_pool.ptr) == int(pool_o["ptr"])
        assert abs(st_g.optimizer.param_groups[0]['lr'] - st_o["lr"]) < 1e-12
        _check_state(st_g, st_o, it, before)
        _sync_from_oracle(st_g, st_o, bank_o, ptr_o)
    if C >= 4:
        assert keys_seen > 0                   # C = 4: the 5-D enqueue ran inside the step on both sides
    print(f"worst parameter-update deviation so far: L2 {_STATS['e2']:.3f} (bound 0.1), element-wise {_STATS['emax']:.3f} (bound 0.3)")


@pytest.mark.parametrize("side_mode", [2, 3])
def test_pass_concurrency_3d_equals_the_single_stream_step(side_mode):
    """(side_mode 3: the teacher's FeatureExtractor behind its pass on the second stream, the warped pass started as soon as the host
    has drawn the warp instead of behind the main stream's queue, bank appends queued after it.)  The 3-D step with the teacher's grouped pass and the gradient-free warped pass on the second stream (train_arco_3d.PASS_SIDE = 2,
    the default) against the single-stream step (0), four steps from equal state: loss terms, weights, BatchNorm buffers."""
    from arco_amd import ops, train_arco_3d as T3
    prev = T3.PASS_SIDE
    try:
        sts = []
        for mode in (0, side_mode):
            T3.PASS_SIDE = mode
            random.seed(3); np.random.seed(3); torch.manual_seed(3)
            args = T3.build_parser().parse_args(["--batch_size", "1", "--queue_size", "200", "--synthetic", "1", "--num_classes", "4",
                                                 "--num_queries", "48", "--num_negatives", "16", "--k1", "1.0"])
            args.patch_size = [32, 32, 32]
            st = T3.ArcoStep3D(args, "cuda:0")
            for m in (st.model, st.ema_model):
                _drop_off(m)
            sts.append(st)
        st_a, st_b = sts

        def sync(dst, src):       # every step starts from EQUAL state (two fp32 trajectories drift apart through the decisions)
            with torch.no_grad():
                dst.optimizer.flat_p.copy_(src.optimizer.flat_p)
                dst.optimizer.flat_buf.copy_(src.optimizer.flat_buf)
                dst.optimizer._started = list(src.optimizer._started)
                for g_d, g_s in zip(dst.optimizer.param_groups, src.optimizer.param_groups):
                    g_d['lr'] = g_s['lr']
                for md, ms in ((dst.model, src.model), (dst.ema_model, src.ema_model), (dst.k_feature_extractor, src.k_feature_extractor)):
                    for (kd, vd), (ks, vs) in zip(md.state_dict().items(), ms.state_dict().items()):
                        vd.copy_(vs)
                dst.memobank = [[m[0].clone()] for m in src.memobank]
                dst.queue_ptrlis = [q.clone() if torch.is_tensor(q) else q for q in src.queue_ptrlis]
            dst.iter_num = src.iter_num
            ops.bump_weight_epoch()

        for it in range(4):
            l, ll = T3.synthetic_volume_batch(1, (32, 32, 32), 4, 10 + it, "cuda:0")
            u, _ = T3.synthetic_volume_batch(1, (32, 32, 32), 4, 20 + it, "cuda:0")
            sync(st_b, st_a)
            terms = []
            for st, mode in ((st_a, 0), (st_b, side_mode)):       # PASS_SIDE is read at step time: the parametrized mode, not a constant
                T3.PASS_SIDE = mode
                random.seed(50 + it); np.random.seed(50 + it); torch.manual_seed(50 + it)
                st.step(l, ll, u)
                torch.cuda.synchronize()
                terms.append({k: float(v) for k, v in st.last_terms.items()})
            for k in terms[0]:
                np.testing.assert_allclose(terms[1][k], terms[0][k], rtol=1e-4, atol=1e-6, err_msg=f"step {it} {k}")
            pa, pb = st_a.optimizer.flat_p, st_b.optimizer.flat_p
            assert float((pa - pb).abs().max()) <= 1e-5 * float(pa.abs().max()), it
        for (k, va), (_, vb) in zip(st_a.model.state_dict().items(), st_b.model.state_dict().items()):
            if va.is_floating_point() and "running" in k:
                np.testing.assert_allclose(vb.cpu().numpy(), va.cpu().numpy(), rtol=1e-3, atol=1e-5, err_msg=k)
    finally:
        T3.PASS_SIDE = prev


@pytest.mark.parametrize("graphs", [0, 1])
def test_weight_gradients_on_the_side_stream_3d_equal_the_single_stream_step(graphs):
    """ops.WGRAD_SIDE = 3 (the 3-D trainer's default since round 6: every layer's weight gradient forks behind its data gradient and
    runs beside the next layer's BatchNorm backward; joined before the optimiser) against 0 (everything on one stream), three steps
    from equal state, eager and graph-replayed student passes: the same kernels on the same operands - weights bit-identical."""
    from arco_amd import ops, train_arco_3d as T3
    prev = ops.WGRAD_SIDE
    try:
        sts = []
        for mode in (0, 3):
            random.seed(4); np.random.seed(4); torch.manual_seed(4)
            args = T3.build_parser().parse_args(["--batch_size", "1", "--queue_size", "200", "--synthetic", "1", "--num_classes", "4",
                                                 "--num_queries", "48", "--num_negatives", "16", "--k1", "1.0", "--graph_train", str(graphs)])
            args.patch_size = [32, 32, 32]
            st = T3.ArcoStep3D(args, "cuda:0")
            assert ops.WGRAD_SIDE == 3                     # the constructor's default (ARCO_WGRAD_SIDE unset)
            for m in (st.model, st.ema_model):
                _drop_off(m)
            sts.append(st)
        st_a, st_b = sts
        with torch.no_grad():
            st_b.optimizer.flat_p.copy_(st_a.optimizer.flat_p)
            for md, ms in ((st_b.model, st_a.model), (st_b.ema_model, st_a.ema_model), (st_b.k_feature_extractor, st_a.k_feature_extractor)):
                for (kd, vd), (ks, vs) in zip(md.state_dict().items(), ms.state_dict().items()):
                    vd.copy_(vs)
        ops.bump_weight_epoch()
        for it in range(3):
            l, ll = T3.synthetic_volume_batch(1, (32, 32, 32), 4, 10 + it, "cuda:0")
            u, _ = T3.synthetic_volume_batch(1, (32, 32, 32), 4, 20 + it, "cuda:0")
            for st, mode in ((st_a, 0), (st_b, 3)):
                ops.WGRAD_SIDE = mode
                random.seed(60 + it); np.random.seed(60 + it); torch.manual_seed(60 + it)
                st.step(l, ll, u)
                torch.cuda.synchronize()
            assert torch.equal(st_a.optimizer.flat_p, st_b.optimizer.flat_p), it
    finally:
        ops.WGRAD_SIDE = prev


def test_three_steps_3d_free_running_chain_is_reported():
    """The same three 3-D steps as `sparse_c4` above, FREE-RUNNING: the HIP step and the CPU oracle start from equal
    state once and are never re-synchronised, and the oracle takes its own gradient-free decisions (no `force=`).  The docs argue that
    such a chain cannot be held to 1e-3 (V-Net gradients are conditioned like ReLU-flip counts, one flipped threshold voxel shifts
    every later draw of the host generator); this test turns the argument into numbers: per step the relative deviation of every loss
    term, whether the two sides still draw the same samples, and the largest parameter deviation - printed, written to
    gpurun_out/r06_free_chain3d.json when that directory exists, and held only to loose sanity bounds (supervised / unsupervised
    terms within 5 %, everything finite).  The re-synced variants above stay the 1e-3 parity gate."""
    import json
    import os
    from arco_amd import ops, train_arco_3d as T3
    C, b, patch, Q, Nn, qs, lr = 4, 2, (32, 32, 32), 48, 16, 200, 0.01
    vnet_sd = _state(C)
    fe_sd = fx.fe_state(61, FEA, 16, nd=3)
    qrep_w = [_qrep_w(71), _qrep_w(72)]
    argv = ["--batch_size", str(b), "--queue_size", str(qs), "--synthetic", "1", "--num_classes", str(C), "--num_queries", str(Q),
            "--num_negatives", str(Nn), "--k1", "1.0", "--base_lr", str(lr), "--dense_head", "0", "--eqv_pass", "1", "--apply_aug", "cutmix",
            "--func", "asmc", "--graphs", "0", "--strong_threshold", "0.3"]
    args = T3.build_parser().parse_args(argv)
    args.patch_size = list(patch)
    random.seed(3); np.random.seed(3); torch.manual_seed(3)
    st_g = T3.ArcoStep3D(args, "cuda:0")
    st_g.model.load_state_dict(vnet_sd, strict=True)
    st_g.ema_model.load_state_dict(vnet_sd, strict=True)
    st_g.q_feature_extractor.load_state_dict(fe_sd, strict=True)
    st_g.k_feature_extractor.load_state_dict(fe_sd, strict=True)
    with torch.no_grad():
        st_g.q_representation[0].weight.copy_(qrep_w[0])
        st_g.q_representation[1].weight.copy_(qrep_w[1])
    for m in (st_g.model, st_g.ema_model):
        _drop_off(m)
    ops.bump_weight_epoch()
    st_o = cpu_step3d.make_state(vnet_sd, fe_sd, qrep_w, base_lr=lr, max_iterations=args.max_iterations)
    bank_o = [[m[0].detach().cpu().clone()] for m in st_g.memobank]
    ptr_o = [torch.zeros(1, dtype=torch.long) for _ in range(C)]
    qsz = list(st_g.queue_size)
    rs = np.random.RandomState(13)

    def gens():
        ns = np.random.get_state()
        return (random.getstate(), ns[1].tobytes(), ns[2:], torch.get_rng_state().numpy().tobytes())

    random.seed(10); np.random.seed(10); torch.manual_seed(10)
    g_state = o_state = (random.getstate(), np.random.get_state(), torch.get_rng_state())
    report = []
    for it in range(3):
        l, lab, u = _volumes(rs, b, patch, C)
        random.setstate(g_state[0]); np.random.set_state(g_state[1]); torch.set_rng_state(g_state[2])
        st_g.step(l.cuda(), lab.cuda(), u.cuda())
        g_state, g_cmp = (random.getstate(), np.random.get_state(), torch.get_rng_state()), gens()
        random.setstate(o_state[0]); np.random.set_state(o_state[1]); torch.set_rng_state(o_state[2])
        cpu_step3d.step(st_o, l, lab, u, bank_o, ptr_o, qsz, n_cls=C, k1=1.0, k3=args.k3, k4=args.k4, delta_n=args.strong_threshold_u2pl,
                        strong_threshold=args.strong_threshold, weak_threshold=args.weak_threshold, func="asmc", nq=Q, nn_=Nn,
                        tps_sigma=args.tps_sigma, apply_aug="cutmix", eqv_pass=True, pool=None, topk=5, force=None)
        o_state, o_cmp = (random.getstate(), np.random.get_state(), torch.get_rng_state()), gens()
        to, tg = st_o["last_terms"], st_g.last_terms
        rel = {k: abs(float(tg[k]) - to[k]) / max(abs(to[k]), 1e-6) for k in ("ce", "dice", "unsup", "reco", "eqv")}
        sd_g = st_g.model.state_dict()
        wdev = max(float((sd_g[k].cpu() - v.detach()).abs().max()) / max(1e-6, float(v.detach().abs().max()))
                   for k, v in st_o["student"].items() if v.requires_grad)
        report.append(dict(step=it, rel_dev=rel, same_host_draws=bool(g_cmp == o_cmp), max_rel_param_dev=wdev,
                           bank_lengths_equal=[int(x[0].shape[0]) for x in bank_o] == [int(m[0].shape[0]) for m in st_g.memobank]))
        for k in ("ce", "dice", "unsup"):
            assert rel[k] < 5e-2, (it, k, rel)
        assert all(np.isfinite(list(rel.values())))
    print("free-running 3-D chain, HIP vs CPU oracle:")
    for r in report:
        print("  ", json.dumps(r))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "r06_free_chain3d.json"), "w") as f:
            json.dump(report, f, indent=1)
