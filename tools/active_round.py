#!/usr/bin/env python
"""One-process-per-GPU shape of the reference's active-learning loop (active_train.py:82-85,440-527) on the HIP path.

    python tools/active_round.py [--rounds 2 --steps 4 ...]                                   # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tools/active_round.py --rounds 2 ...                                  # N GPUs, RCCL

What the reference does with nn.DataParallel + patch_replication_callback and what this driver does instead:

  reference (one process, N replicas)                         here (N processes, one GPU each)
  ---------------------------------------------------------   --------------------------------------------------------------
  model = DataParallel(model); patch_replication_callback     model = ModuleWrapper(model)  (keeps `.module` for the selectors,
                                                              core_set.py:44,52); gradients averaged by GradientAverager
                                                              (bucketed all-reduce overlapped with backward)
  loss on the gathered logits, / global batch (loss.py:39-51)  SegmentationLosses(global_batch=True): numerator / valid count /
                                                              batch size exchanged in one 3-float all-reduce
  SynchronizedBatchNorm2d through the replication callback    --sync-bn: SynchronizedBatchNorm2d all-reduces its [2K] sums
                                                              (models/sync_batchnorm, batchnorm.py:113-125); else per-GPU BN and
                                                              the buffers of rank 0 are broadcast before every scoring pass
                                                              (DataParallel scores with device 0's buffers)
  selector(model, unlabeled keys, k) on one process           the same call on every rank: each scores a contiguous shard of the
                                                              key list, scores are all-gathered, every rank sorts -> same picks
  training_set.expand_training_set(selected)                  labelled += selected; unlabeled -= selected (every rank alike)

The pool is synthetic (per-key seeded tensors): no dataset, no checkpoints, no tensorboard -- the data layer
and the savers of active_train.py are outside the hot path.  Rank 0 prints one JSON line; with DASS_ROUND_DUMP=<dir> every
rank also writes its selections and a checksum of its parameters there (the world-2 test compares them).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="mobilenet")
    ap.add_argument("--classes", type=int, default=19)
    ap.add_argument("--size", type=int, default=129)
    ap.add_argument("--pool", type=int, default=24, help="images in the synthetic pool")
    ap.add_argument("--seed-set", type=int, default=4, help="initially labelled images")
    ap.add_argument("--select", type=int, default=4, help="images added per round (--active-batch-size)")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=4, help="train steps per round")
    ap.add_argument("--batch", type=int, default=2, help="per-GPU batch")
    ap.add_argument("--mode", default="mc_dropout", choices=["mc_dropout", "ceal_entropy", "coreset"])
    ap.add_argument("--mc-steps", type=int, default=4)
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--lr", type=float, default=0.01)
    return ap.parse_args()


def sample(key, size, classes):
    """seeded synthetic (image, label) of pool key b'img_%06d' -- content independent of the sharding"""
    idx = int(key.decode("ascii").split("_")[1])
    g = torch.Generator().manual_seed(1000 + idx)
    img = torch.randn(3, size, size, generator=g)
    lab = torch.randint(0, classes, (size, size), generator=g).float()
    lab[: size // 10] = 255
    return img, lab


def main():
    args = parse()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("DASS_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("tools/active_round.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("DASS_BENCH_BACKEND", "nccl"))  # nccl = RCCL on ROCm
    import constants
    from active_selection import get_active_selection_class  # noqa: F401  (the factory is what active_train.py:445 calls)
    from active_selection.ceal import ActiveSelectionCEAL
    from active_selection.core_set import ActiveSelectionCoreSet
    from active_selection.mc_dropout import ActiveSelectionMCDropout
    from dass_hip.dist import GradientAverager, ModuleWrapper
    from dass_hip.optim import SGD
    from models.deeplab import DeepLab
    from utils.loss import SegmentationLosses

    torch.manual_seed(1234)  # identical initial weights on every rank
    core = DeepLab(backbone=args.backbone, output_stride=16, num_classes=args.classes, sync_bn=args.sync_bn, freeze_bn=False,
                   pretrained=False).to(dev)
    optimizer = SGD([{"params": core.get_1x_lr_params(), "lr": args.lr}, {"params": core.get_10x_lr_params(), "lr": args.lr * 10}],
                    momentum=0.9, weight_decay=5e-4, nesterov=False)     # taken BEFORE wrapping, as active_train.py:49-50
    params = [p for g in optimizer.param_groups for p in g["params"]]
    model = ModuleWrapper(core)
    averager = GradientAverager(params)
    criterion = SegmentationLosses(cuda=True, global_batch=True).build_loss("ce")
    keys = [("img_%06d" % i).encode("ascii") for i in range(args.pool)]
    labelled, unlabeled = keys[: args.seed_set], keys[args.seed_set:]

    def loader_factory(images, include_labels, bs=args.batch):
        for i in range(0, len(images), bs):
            pairs = [sample(k, args.size, args.classes) for k in images[i:i + bs]]
            img = torch.stack([p[0] for p in pairs])
            yield {"image": img, "label": torch.stack([p[1] for p in pairs])} if include_labels else img

    def broadcast_buffers():
        if dist is not None and not args.sync_bn:   # DataParallel scores with the buffers of device 0
            for b in core.buffers():
                dist.broadcast(b, src=0)

    constants.MC_STEPS = args.mc_steps
    score_log = []
    if args.mode == "mc_dropout":
        selector = ActiveSelectionMCDropout(args.classes, None, args.size, args.batch, loader_factory=loader_factory)
        pick = lambda cand, k: list(selector.get_vote_entropy_for_images(model, cand, k))  # noqa: E731
    elif args.mode == "ceal_entropy":
        selector = ActiveSelectionCEAL(args.classes, None, args.size, args.batch, loader_factory=loader_factory)

        def pick(cand, k):
            chosen, entropies = selector.get_maximum_entropy_samples(model, cand, k)
            score_log.append([round(float(e), 7) for e in entropies])   # (this selector returns its scores: kept for the tests)
            return list(chosen)
    else:
        selector = ActiveSelectionCoreSet(None, args.size, args.batch, loader_factory=loader_factory)
        pick = lambda cand, k: selector.get_k_center_greedy_selections(k, model, cand, list(labelled))  # noqa: E731

    history, losses, t0 = [], [], time.perf_counter()
    score_log.clear()
    for rnd in range(args.rounds):
        # ---- train: every step consumes a GLOBAL batch of world * batch labelled keys; rank r takes its slice
        model.train()
        gb = world * args.batch
        for step in range(args.steps):
            first = (rnd * args.steps + step) * gb
            mine = [labelled[(first + rank * args.batch + j) % len(labelled)] for j in range(args.batch)]
            pairs = [sample(k, args.size, args.classes) for k in mine]
            x = torch.stack([p[0] for p in pairs]).to(dev)
            y = torch.stack([p[1] for p in pairs]).to(dev)
            # Dropout2d masks per (step, position in the GLOBAL batch) from their own seeds: what an image sees
            # does not depend on how the batch is cut over the ranks
            m1, m2 = [], []
            for j in range(args.batch):
                gm = torch.Generator().manual_seed(7_000_000 + (rnd * args.steps + step) * 1000 + rank * args.batch + j)
                m1.append((torch.rand(256, generator=gm) >= 0.5).float() * 2.0)
                m2.append((torch.rand(256, generator=gm) >= constants.MC_DROPOUT_RATE).float() / (1.0 - constants.MC_DROPOUT_RATE))
            optimizer.zero_grad(set_to_none=True)
            loss = criterion(model(x, dropout_masks=(torch.stack(m1).to(dev), torch.stack(m2).to(dev))), y)
            loss.backward()
            averager.finish()
            optimizer.step()
            losses.append(float(loss.detach()))
        # ---- score the unlabeled pool (sharded over the ranks) and move the selection into the labelled set
        model.eval()
        broadcast_buffers()
        chosen = pick(list(unlabeled), min(args.select, len(unlabeled)))
        history.append([k.decode("ascii") for k in chosen])
        labelled = labelled + list(chosen)
        unlabeled = [k for k in unlabeled if k not in set(chosen)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h = hashlib.sha256()
    for p in core.parameters():
        h.update(p.detach().float().cpu().numpy().tobytes())
    out = {"rank": rank, "world": world, "mode": args.mode, "sync_bn": bool(args.sync_bn), "selections": history,
           "labelled": len(labelled), "losses": [round(v, 6) for v in losses], "scores": score_log, "param_sha256": h.hexdigest(), "seconds": round(dt, 3)}
    dump = os.environ.get("DASS_ROUND_DUMP")
    if dump:
        with open(os.path.join(dump, "rank%d.json" % rank), "w") as f:
            json.dump(out, f)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
