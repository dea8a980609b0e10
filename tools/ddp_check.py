"""Run by tests/test_dist_gpu.py under torchrun (2 ranks sharing one GPU over gloo): after a few data-parallel
steps every rank must hold bit-identical banks, queue pointers, student parameters and teacher parameters."""
import os, sys, hashlib
os.environ.setdefault("OMP_NUM_THREADS", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import random, numpy as np, torch
import torch.distributed as td
from arco_amd import dist as adist, train_arco_2d as T

rank, world = adist.init()
assert world == 2 and adist.is_dist()
dev = torch.device("cuda", adist.local_rank())
torch.cuda.set_device(dev)
# entropy percentiles of the GLOBAL batch: the phased radix select with summed histograms must give
# the masks that np.percentile over both ranks' valid entropies gives
from arco_amd import glue, _lib as L_
g = torch.Generator().manual_seed(100 + rank)
pred = (3 * torch.randn(3, 4, 24, 40, generator=g)).to(dev)
lab_l = torch.randint(0, 4, (3, 24, 40), generator=g).to(dev)
lab_u = torch.randint(-1, 4, (3, 24, 40), generator=g).to(dev)
assert glue.state_reduce_hook is adist.allreduce_sum
low, high = glue.entropy_masks(pred, lab_l, lab_u, 20.0)
r_, ld_, b_, C_n, P_ = glue._geom(pred)
ent = torch.empty(b_ * P_, dtype=torch.float32, device=dev)
L_.call("arco_softmax_rows", L_.ptr(r_), ld_, b_ * P_, C_n, P_, None, None, None, L_.ptr(ent))
ents, labs = [torch.empty_like(ent) for _ in range(world)], [torch.empty_like(lab_u) for _ in range(world)]
td.all_gather(ents, ent); td.all_gather(labs, lab_u.contiguous())
vals = torch.cat([e[l.reshape(-1) >= 0] for e, l in zip(ents, labs)]).cpu().numpy()
t_lo, t_hi = np.float32(np.percentile(vals, 20.0)), np.float32(np.percentile(vals, 80.0))
valid = (lab_u.reshape(-1) >= 0)
exp_low = ((ent <= float(t_lo)) & valid).float().view(3, 1, 24, 40)
exp_high = ((ent >= float(t_hi)) & valid).float().view(3, 1, 24, 40)
assert torch.equal(low[3:], exp_low) and torch.equal(high[3:], exp_high), "global entropy thresholds differ"
assert torch.equal(low[:3, 0], (lab_l >= 0).float())
local_lo = np.float32(np.percentile(ent[valid].cpu().numpy(), 20.0))
assert local_lo != t_lo                                           # the per-rank threshold would have been different

# ---- SURVEY 8e parity definition (ii) on the product path: with num_queries / world anchors per rank, global prototypes and
# rank-ordered key gathers, the rank-averaged loss equals the single-process loss on the CONCATENATED batch when the single
# process replays the ranks' sampled indices; banks and pointers bit-identical (tests/test_dist_parity_cpu.py is the CPU twin)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fixture_inputs as fx
from arco_amd import _contrast as C_
Cn, Dn, Qn, Nn_, QSn, Bn_, SPn = 4, 32, 64, 16, 160, 2, (24, 24)
inp = {k: v.to(dev) for k, v in fx.loss_inputs(50 + rank, b=Bn_, n_cls=Cn, feat=Dn, spatial=SPn).items()}
rep = inp["rep"].clone().requires_grad_(True)
bank, ptr, qs = fx.fresh_bank(Cn, Dn, QSn, 'zeros')
q_rank = adist.anchors_for_rank(Qn, "split")
assert q_rank == Qn // world
traces = []
for step in range(2):
    torch.manual_seed(1000 * step + rank)                        # rank-seeded sampler sequence
    pl = C_.contrast_masks(inp["label_l"], inp["label_u"], inp["prob_l"], inp["prob_u"], inp["low_mask"], inp["high_mask"], 0.97)
    C_.contrast_counts(pl, bank, qs, q_rank, Nn_)
    tr = {}
    C_.contrast_draw(pl, 'smc', _trace=tr)
    C_.contrast_enqueue(pl, inp["rep_teacher"], bank, ptr, qs, _trace=tr)
    loss_r, _ = C_.contrast_infonce(pl, C_.GatherRowsFn.apply(rep, pl.anchor_pix), bank)
    assert pl.valid_seg == Cn and len(pl.entries) == Cn
    traces.append([(pl.lists[k][a].long().cpu(), n.cpu()) for (k, vc, a, n) in pl.entries])
loss_r.backward()
l_dp = loss_r.detach().clone()
td.all_reduce(l_dp); l_dp /= world
every = [None] * world
td.all_gather_object(every, dict(inp={k: v.cpu() for k, v in inp.items()}, traces=traces))
hooks = (C_.key_gather_hook, C_.count_gather_hook, C_.tail_gather_hook, C_.proto_reduce_hook, C_.tail_gather_all_hook,
         C_.totals_gather_hook, C_.count_table_hook)
C_.key_gather_hook = C_.count_gather_hook = C_.tail_gather_hook = C_.proto_reduce_hook = C_.tail_gather_all_hook = None     # single-process semantics
C_.totals_gather_hook = C_.count_table_hook = None
try:
    halves = lambda key: torch.cat([e["inp"][key][:Bn_] for e in every] + [e["inp"][key][Bn_:] for e in every]).to(dev)
    cat = lambda key: torch.cat([e["inp"][key] for e in every]).to(dev)
    rep_all = halves("rep").requires_grad_(True)
    P = SPn[0] * SPn[1]
    to_global = lambda rows, r: torch.where(rows < Bn_ * P, rows + r * Bn_ * P, rows - Bn_ * P + world * Bn_ * P + r * Bn_ * P)
    bank1, ptr1, qs1 = fx.fresh_bank(Cn, Dn, QSn, 'zeros')
    for step in range(2):
        pl1 = C_.contrast_masks(cat("label_l"), cat("label_u"), cat("prob_l"), cat("prob_u"), halves("low_mask"), halves("high_mask"), 0.97)
        C_.contrast_counts(pl1, bank1, qs1, Qn, Nn_)
        C_.contrast_enqueue(pl1, halves("rep_teacher"), bank1, ptr1, qs1, defer_anchor_pix=True)
        pl1.entries = []
        for k in range(pl1.valid_seg):
            g_pix = torch.cat([to_global(every[r]["traces"][step][k][0], r) for r in range(world)]).to(dev)
            cand = pl1.lists[k][:int(pl1.n_anchor[k])].long()
            pos = torch.searchsorted(cand, g_pix)
            assert torch.equal(cand[pos], g_pix)
            pl1.entries.append((k, pl1.valid_classes[k], pos, torch.cat([every[r]["traces"][step][k][1] for r in range(world)]).to(dev)))
        C_.contrast_anchor_pix(pl1)
        l_single, _ = C_.contrast_infonce(pl1, C_.GatherRowsFn.apply(rep_all, pl1.anchor_pix), bank1)
finally:
    (C_.key_gather_hook, C_.count_gather_hook, C_.tail_gather_hook, C_.proto_reduce_hook, C_.tail_gather_all_hook,
     C_.totals_gather_hook, C_.count_table_hook) = hooks
assert abs(float(l_single) - float(l_dp)) < 1e-5 * max(1.0, abs(float(l_single))), (float(l_single), float(l_dp))
for c in range(Cn):
    assert torch.equal(bank[c][0], bank1[c][0]) and int(ptr[c]) == int(ptr1[c]), c
l_single.backward()
g_all = rep_all.grad
mine = torch.cat((g_all[rank * Bn_:(rank + 1) * Bn_], g_all[world * Bn_ + rank * Bn_: world * Bn_ + (rank + 1) * Bn_]))
assert torch.allclose(rep.grad / world, mine, rtol=1e-3, atol=1e-8), float((rep.grad / world - mine).abs().max())
td.barrier()

random.seed(7); np.random.seed(7); torch.manual_seed(7)          # same sampler sequence on every rank
args = T.build_parser().parse_args(["--batch_size", "2", "--queue_size", "200", "--synthetic", "1", "--num_queries", "64",
                                    "--num_negatives", "32", "--k1", "1.0", "--base_lr", "0.05"])
args.patch_size = [64, 64]
st = T.ArcoStep2D(args, dev)
assert args.graph_train == 1                 # trainer default: the student passes replay as HIP graphs from their third call
fired = []
real_async = adist.allreduce_bucket_async
def counting_async(opt, start):
    fired.append(len(adist._pending_buckets) == 0)
    return real_async(opt, start)
adist.allreduce_bucket_async = counting_async
real_draw = C_.contrast_draw
N_STEPS, ZERO_STEP = 8, 5
for i in range(N_STEPS):
    l_img, l_lab = T.synthetic_batch(2, args.patch_size, 4, 100 + 10 * i + rank, dev)      # different data per rank
    u_img, _ = T.synthetic_batch(2, args.patch_size, 4, 200 + 10 * i + rank, dev)
    if i == ZERO_STEP and rank == 1:
        # ADVICE r3 (high), on the real step: THIS rank's loss takes the degenerate-batch path (no entries -> `weight.sum() * 0`,
        # the heads' marker never fires) while rank 0's does not - the gradient exchange must still be the same two collectives
        def no_entries(pl, *a_, **k_):
            r = real_draw(pl, *a_, **k_)
            C_.contrast_draw_finish(pl)
            pl.entries = []
            return r
        C_.contrast_draw = no_entries
    n_before = len(fired)
    loss, reco = st.step(l_img, l_lab, u_img)
    C_.contrast_draw = real_draw
    if i == ZERO_STEP and rank == 1:
        assert len(fired) == n_before and float(reco) == 0.0, "the zero path must not reach the heads' marker"
    else:
        assert len(fired) == n_before + 1 and fired[-1], f"step {i}: the heads' bucket did not start from inside the backward"
    assert adist._pending_buckets == [] and st.optimizer._bucket_start is None
torch.cuda.synchronize()
assert st.s_train_lu.captured and st.s_train_lu.calls >= N_STEPS       # the async bucket met graph-replayed backward passes (steps 3..)


def digest(ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


mine = [digest([b[0] for b in st.memobank]), digest([p for p in st.queue_ptrlis]), digest(st.optimizer.params),
        digest(list(st.ema_model.parameters())), str([int(b[0].shape[0]) for b in st.memobank])]
both = [None, None]
td.all_gather_object(both, mine)
assert both[0] == both[1], (both[0], both[1])
assert sum(int(b[0].shape[0]) for b in st.memobank) > 4, "banks never grew"
if rank == 0:
    print(f"DDP_OK {N_STEPS} steps, graph-replayed student passes from step 3, heads' bucket fired {len(fired)} times on rank 0, rank 1 took the zero path at step {ZERO_STEP};")
    print("DDP_OK global entropy thresholds; rank-averaged loss == single-process loss on the concatenated batch (%.6f);" % float(l_dp), " banks/ptr/params/teacher identical on 2 ranks; bank lens", mine[4], "loss", float(reco))
