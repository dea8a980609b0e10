"""Stage-1 pre-training on the GPU (arco_amd/pretrain_2D.py, ISD.forward of arco_amd/model_2D.py) against the reference's
golden vectors (g13, two chained iterations) and the CPU oracle; the trainer end to end on synthetic data; the batch
augmentations of transform_student (-m gpu)."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import fixture_inputs as fx        # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = {2: np.load(os.path.join(ROOT, "tests", "golden", "g13_pretrain.npz")),
        3: np.load(os.path.join(ROOT, "tests", "golden", "g14_pretrain3d.npz"))}
G = GOLD[2]


def _stepper(cfg, nd=2):
    from arco_amd import pretrain_2D as P, pretrain_3D as P3
    mod = P if nd == 2 else P3
    args = mod.build_parser().parse_args(["--K", str(cfg["K"]), "--T_s", str(cfg["Ts"]), "--T_t", str(cfg["Tt"]),
                                          "--latent_feature_size", str(cfg["latent_feature_size"]), "--num_classes", str(cfg["num_classes"]),
                                          "--output_pooling_size", str(cfg["output_pooling_size"]), "--cut_size", str(cfg["patch_size"]),
                                          "--labeled_bs", str(cfg["labeled_bs"]), "--batch_size", str(cfg["b"]),
                                          "--base_lr", str(cfg["lr"]), "--max_iterations", "1000000000"])
    args.head_patch = cfg["patch_size"]
    st = (P.PretrainStep2D if nd == 2 else P3.PretrainStep3D)(args, torch.device("cuda", 0))
    heads = fx.isd_head_state(33) if nd == 2 else fx.isd3d_head_state(33)
    if nd == 3:       # the constructor hard-wires 700 patches (112 x 112 x 80 volumes); the test volume has 27
        st.model.queue_mask = torch.zeros_like(heads["queue_mask"]).cuda()
    with torch.no_grad():
        for sub, seed in ((st.model.model, 31), (st.model.ema_model, 32)):
            sd = sub.state_dict()
            for k, v in (fx.unet_state if nd == 2 else fx.vnet_state)(seed).items():
                sd[k].copy_(v)
        sd = st.model.state_dict()
        for k, v in heads.items():
            sd[k].copy_(v)
    for m in st.model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout3d)):
            m.p = 0.0
    st.optimizer._weights_changed()
    return st


@pytest.mark.parametrize("nd", [2, 3])
def test_stage1_two_iterations_vs_reference_golden(nd):
    cfg = fx.STAGE1_CFG if nd == 2 else fx.STAGE1_CFG_3D
    G = GOLD[nd]
    st = _stepper(cfg, nd)
    sd0 = {k: v.clone() for k, v in st.model.state_dict().items()}
    bn1 = "encoder.in_conv.conv_conv.1" if nd == 2 else "block_one.conv.1"
    for it in range(cfg["steps"]):
        im_q, im_k, lab = (fx.stage1_batch if nd == 2 else fx.stage1_batch_3d)(40, it)
        torch.manual_seed(100 + it)
        if it == 0:        # forward alone first (gradients of iteration 0 are in the golden file)
            st.model.zero_grad()
        st.step(im_q.cuda(), lab.cuda(), im_k.cuda())
        t = st.last_terms
        got = [float(t[k]) for k in ("loss", "ce", "dice", "latent", "output")]
        np.testing.assert_allclose(got, G[f"s{it}_terms"], rtol=1e-3, err_msg=f"iteration {it}")
    sd = st.model.state_dict()
    assert int(sd["queue_ptr"]) == int(G["final::queue_ptr"][0]) and int(sd["mask_queue_ptr"]) == int(G["final::mask_queue_ptr"][0])
    np.testing.assert_allclose(sd["queue"].cpu().numpy(), G["final::queue"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(sd["queue_mask"].cpu().numpy(), G["final::queue_mask"], rtol=1e-3, atol=1e-4)
    assert int(sd[f"ema_model.{bn1}.num_batches_tracked"]) == int(G[f"final::ema_model.{bn1}.num_batches_tracked"]) == 4
    for pre in ("model", "ema_model"):
        for n, ref_abs in zip((str(x) for x in G[f"final_{pre}.names"]), G[f"final_{pre}.abs"]):
            v = sd[n].float()
            np.testing.assert_allclose(float(v.double().abs().sum()), ref_abs, rtol=1e-3, atol=1e-5, err_msg=n)
    changed = 0
    for k in G.files:
        if not k.startswith("final::") or k.split("::")[1].startswith(("queue", "mask_queue")):
            continue
        name = k.split("::")[1]
        v = sd[name].detach().cpu()
        if v.dtype == torch.long:
            continue
        changed += int(not torch.equal(v, sd0[name].cpu()))
        v = v if v.numel() <= 20000 else v[::4, ::4]
        # relative to the tensor's scale, as tests/test_step_parity_gpu.py does for updated weights (gradients through
        # train-mode BatchNorm over 4 x 64 x 64 pixels are ill-conditioned element by element)
        err = float(np.abs(v.numpy() - G[k]).max()) / max(1e-6, float(np.abs(G[k]).max()))
        assert err < 2e-3, (name, err)
    assert changed >= 20          # heads (query AND key), predictors, the pinned network tensors all moved


def test_stage1_forward_and_gradients_vs_reference_golden():
    """Iteration 0 in detail: the six outputs of ISD.forward and the gradients (the key heads receive gradients through
    the KLD targets, like the reference)."""
    from arco_amd import glue, pretrain_2D as P
    cfg = fx.STAGE1_CFG
    st = _stepper(cfg)
    im_q, im_k, lab = fx.stage1_batch(40, 0)
    torch.manual_seed(100)
    outputs, ema_output, ema_ll, ll, ema_ol, ol = st.model(im_q.cuda(), im_k.cuda())
    np.testing.assert_allclose(outputs.detach()[:, :, ::2, ::2].cpu().numpy(), G["s0_outputs"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(ema_output.detach()[:, :, ::2, ::2].cpu().numpy(), G["s0_ema_output"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(ema_ll.detach().cpu().numpy(), G["s0_ema_latent_logits"], rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(ll.detach().cpu().numpy(), G["s0_latent_logits"], rtol=1e-3, atol=2e-3)
    for tag, t in (("ema_output_logits", ema_ol), ("output_logits", ol)):
        assert tuple(t.shape) == tuple(G[f"s0_{tag}_shape"])
        np.testing.assert_allclose(t.detach()[::7, ::13].cpu().numpy(), G[f"s0_{tag}_sample"], rtol=1e-3, atol=3e-3)
    ce, dice = glue.supervised_loss(outputs[:cfg["labeled_bs"]], lab[:cfg["labeled_bs"]].cuda().long())
    kld = P.KLD()
    loss = (dice + ce) + kld(ll, ema_ll) + kld(ol, ema_ol)
    st.optimizer.zero_grad()
    loss.backward()
    names = [str(n) for n in G["grad_names"]]
    named = dict(st.model.named_parameters())
    seen = 0
    for i, n in enumerate(names):
        g = named[n].grad
        assert g is not None, n
        np.testing.assert_allclose(float(g.double().abs().sum()), G["grad_abs"][i], rtol=5e-3, atol=2e-6 * g.numel(), err_msg=n)
        if "grad::" + n in G.files and g.numel() <= 20000:
            ref = G["grad::" + n]          # relative to the tensor's scale (BatchNorm backward over few pixels)
            assert float(np.abs(g.cpu().numpy() - ref).max()) <= 5e-3 * float(np.abs(ref).max()) + 1e-6, n
            seen += 1
    assert any(n.startswith("k_latent_head") for n in names) and any(n.startswith("k_outputs_head") for n in names)
    assert seen >= 15


def test_pretrain_trainer_runs_and_feeds_stage2(tmp_path):
    """The trainer end to end on synthetic slices (4 iterations at 64x64), checkpoints written in the reference's format
    (iter_<n>.pth / iter_<n>_ema.pth = U-Net state dicts) and loadable by the stage-2 model."""
    from arco_amd import pretrain_2D as P
    from arco_amd.model_2D import create_model
    snap = str(tmp_path)
    argv = ["--synthetic", "1", "--max_iterations", "4", "--save_every", "2", "--batch_size", "4", "--labeled_bs", "2", "--K", "12",
            "--cut_size", "16", "--output_pooling_size", "4", "--latent_feature_size", "32", "--snapshot_path", snap]
    args = P.build_parser().parse_args(argv)
    args.patch_size = [64, 64]
    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)
    assert P.train(args, snap) == "Training Finished!"
    for n in (2, 4):
        for tag in ("", "_ema"):
            sd = torch.load(os.path.join(snap, f"iter_{n}{tag}.pth"), map_location="cpu")
            net = create_model(num_classes=4)
            net.load_state_dict(sd, strict=True)
            assert all(torch.isfinite(v.float()).all() for v in sd.values())


def test_transform_student_semantics():
    """RandomColorJitter edits the shared image tensor in place (student and teacher both see it), RandomNoise returns
    a new, blurred teacher batch: 8-bit Pillow blur of the (possibly jittered) images; generator consumption as the
    reference (numpy gates, torch for the jitter factors, python random for the radius)."""
    from arco_amd import pretrain_2D as P
    from arco_amd.dataloaders.dataset_withAug import jitter_gray_, _jitter_params
    import arco_oracle as orc
    rs = np.random.RandomState(0)
    img = torch.from_numpy(rs.uniform(size=(3, 1, 32, 40)).astype(np.float32)).cuda()
    ts = P.make_transform_student()
    hits = set()
    for seed in range(12):
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        batch = {'image': img.clone(), 'label': torch.zeros(3, 32, 40).cuda()}
        before = batch['image'].clone()
        student, teacher = P.student_teacher_batches(batch, 2, ts)
        # replay the draws
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        exp = before.clone().cpu()
        jit = not (np.random.uniform(low=0, high=1, size=1) > 0.5)
        if jit:
            for j in range(3):
                order, fac = _jitter_params((0.2, 0.2, 0.2, 0.1))
                jitter_gray_(exp[j], order, fac)
        blur = not (np.random.uniform(low=0, high=1, size=1) > 0.5)
        assert student is batch and torch.allclose(student['image'].cpu(), exp, atol=1e-6)
        if blur:
            sigma = random.uniform(0.15, 1.15)
            want = np.stack([orc.gaussian_blur_u8(orc.q8(exp[j].numpy()), sigma)[0] for j in range(3)]).astype(np.float32) / np.float32(255.0)
            assert teacher is not batch and teacher['image'].shape == (3, 32, 40)
            np.testing.assert_allclose(teacher['image'].cpu().numpy(), want, atol=1e-6)
        else:
            assert teacher['image'] is batch['image']        # (the jitter returns a new dict around the same tensor)
        hits.add((jit, blur))
    assert len(hits) == 4


def test_pretrain3d_trainer_runs(tmp_path):
    """pretrain_3D end to end on synthetic volumes (2 iterations at 32^3 with 16-voxel head patches), checkpoints loadable
    by the stage-2 V-Net."""
    from arco_amd import pretrain_3D as P3
    from arco_amd.model_3D import create_model_3d
    snap = str(tmp_path)
    args = P3.build_parser().parse_args(["--synthetic", "1", "--max_iterations", "2", "--save_every", "2", "--K", "4",
                                         "--output_pooling_size", "2", "--latent_feature_size", "16", "--snapshot_path", snap])
    args.patch_size, args.head_patch = [32, 32, 32], 16
    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)
    import arco_amd.model_3D as M3
    real = M3.ISD_3d.__init__

    def small_queue(self, *a, **k):            # 27 patches instead of the hard-wired 700
        real(self, *a, **k)
        self.queue_mask = torch.nn.functional.normalize(torch.randn(self.K, 27, self.num_classes * 8), dim=-1)
    M3.ISD_3d.__init__ = small_queue
    try:
        assert P3.train(args, snap) == "Training Finished!"
    finally:
        M3.ISD_3d.__init__ = real
    for tag in ("", "_ema"):
        sd = torch.load(os.path.join(snap, f"iter_2{tag}.pth"), map_location="cpu")
        create_model_3d(num_classes=2).load_state_dict(sd, strict=True)
        assert all(torch.isfinite(v.float()).all() for v in sd.values())


def test_transform_student_3d_semantics():
    """3-D variants (la_heart.py:254-294): both edit the volume IN PLACE, slice by slice along the last axis; the blur
    stores the 8-bit values 0..255 back (the reference drops the /255 here)."""
    from arco_amd import pretrain_3D as P3
    import arco_oracle as orc
    rs = np.random.RandomState(1)
    vol = torch.from_numpy(rs.uniform(size=(2, 1, 24, 20, 5)).astype(np.float32)).cuda()
    ts = P3.make_transform_student()
    seen = set()
    for seed in range(10):
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        batch = {'image': vol.clone(), 'label': torch.zeros(2, 24, 20, 5).cuda()}
        student, teacher = P3.student_teacher_batches(batch, 1, ts)
        assert student['image'] is batch['image'] and teacher is batch
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        from arco_amd.dataloaders.dataset_withAug import jitter_gray_, _jitter_params
        exp = vol.clone().cpu()
        jit = not (np.random.uniform(low=0, high=1, size=1) > 0.5)
        if jit:
            for j in range(2):
                for t in range(5):
                    order, fac = _jitter_params((0.02, 0.02, 0.02, 0.01))
                    jitter_gray_(exp[j, :, :, :, t], order, fac)
        blur = not (np.random.uniform(low=0, high=1, size=1) > 0.5)
        if blur:
            sigma = random.uniform(0.15, 1.15)
            for j in range(2):
                for t in range(5):
                    exp[j, 0, :, :, t] = torch.from_numpy(orc.gaussian_blur_u8(orc.q8(exp[j, :, :, :, t].numpy()), sigma)[0].astype(np.float32))
        np.testing.assert_allclose(batch['image'].cpu().numpy(), exp.numpy(), atol=1e-5)
        seen.add((jit, blur))
    assert len(seen) >= 3
