"""Kernel-level co-residence stress of every LDS-ring kernel ("Next round" item 1).

For each case -- the four conv_chain_kernel instantiations, conv_kxr_kernel<3,2> / <5,2>, the three-buffer ring of conv_planar_kernel (256x128
and 128x64 tiles), at their 32-clip and 4-clip shapes -- the kernel is launched N times on ROTATED input sets into NaN-filled outputs while a
second process (scripts/gpu_hammer.py, started before this process touches the GPU) loads the same GPU, and every output is compared bit for
bit with the SOLO launch of the same input set (taken before the hammer starts).  A stale ring stage, a lost store or a stale register
prefetch shows as a differing element; for the chain kernel two KNOWN-ANSWER input sets make the wrong element say where its value came from:

  code-x   the shortcut tensor carries (pixel, channel) codes and conv3's input is forced to zero (b2 = -1e4): y == x bit for bit, a wrong
           element decodes to the (pixel, channel) whose shortcut value was used instead
  code-m   mid1 carries the codes, conv2 = centre-tap identity, conv3 = four copies, conv1' = pick copy 1, x = 0: y[c] == mid1[c % 64],
           z == mid1; a wrong element decodes to the mid1 value (or the weight piece) that reached it

usage: ring_stress.py [--hammer none|matmul|ew|mixed|churn|pipe] [--launches N] [--cases chain,kxr,planar] [--clips 32,4] [--probe]
                      [--json FILE] [--seconds S]
  --probe   the library is a -DCH_PROBE=1 build (STM_LIBRARY=...): reads the per-wave {HW_ID start/end, largest barrier-to-barrier gap} records
            of every chain launch and reports waves that moved (context save / restore) or stalled > 200 us
exit code 0 = every launch of every case bit-equal, 1 = differences (printed), 2 = usage / setup
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hammer", default="matmul")
    ap.add_argument("--launches", type=int, default=200)
    ap.add_argument("--cases", default="chain,kxr,planar")
    ap.add_argument("--clips", default="32,4")
    ap.add_argument("--probe", action="store_true")
    ap.add_argument("--json", default=None)
    ap.add_argument("--seconds", type=float, default=600.0, help="hammer lifetime bound")
    ap.add_argument("--batch", type=int, default=8, help="launches in flight between two host synchronisations")
    ap.add_argument("--sets", type=int, default=3, help="rotated input sets per case")
    return ap.parse_args()


def start_hammer(mode, seconds):
    if mode == "none":
        return None
    d = tempfile.mkdtemp(prefix="stm_hammer_")
    files = {k: os.path.join(d, k) for k in ("ready", "go", "stop")}
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "scripts", "gpu_hammer.py"), mode, str(seconds), "--ready", files["ready"],
                             "--go", files["go"], "--stop", files["stop"]])
    t0 = time.time()
    while not os.path.exists(files["ready"]):
        if proc.poll() is not None:
            raise SystemExit(f"ring_stress: the hammer exited with {proc.returncode} before it was ready")
        if time.time() - t0 > 300:
            proc.kill()
            raise SystemExit("ring_stress: the hammer did not become ready in 300 s")
        time.sleep(0.1)
    return proc, files


def main():
    args = parse()
    hammer = start_hammer(args.hammer, args.seconds)           # before this process touches the GPU
    try:
        return run(args, hammer)
    finally:                                                   # an exception while the cases are prepared must not leave the neighbour (and the pipes it
        if hammer is not None and hammer[0].poll() is None:    # inherited) behind: a caller reading our stdout would wait for the hammer's whole lifetime
            open(hammer[1]["stop"], "w").write("stop\n")
            try:
                hammer[0].wait(timeout=60)
            except subprocess.TimeoutExpired:
                hammer[0].kill()


def run(args, hammer):
    import torch
    from stmask_amd import _lib, ops
    from stmask_amd.planar import PlanarConv

    dev = "cuda"
    torch.cuda.set_device(0)
    H, W = 96, 160
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, scale=1.0: torch.randn(*s, generator=g, device=dev) * scale          # (device generator: the big input sets are made in place)
    geo = _lib.ConvGeom()
    geo.C, geo.Cout, geo.kh, geo.kw, geo.sh, geo.sw, geo.ph, geo.pw, geo.groups, geo.fmt = 64, 64, 3, 3, 1, 1, 1, 1, 1, 1
    ops.planar_range_flag()
    results = []
    dbg = None
    if args.probe:
        L = _lib.lib()
        dbg = torch.zeros(256 * 6 * 4, dtype=torch.int64, device=dev)
        L.stm_debug_chain_timing.restype = None
        L.stm_debug_chain_timing(ctypes.c_void_p(dbg.data_ptr()))

    def coded(n, C, period):
        """fp32 [n, C]: 32768 + ((m % period) * C + c) / 64 -- 22-bit values, exact in two fp16 planes, one code per (pixel mod period, channel)."""
        m = torch.arange(n, dtype=torch.int64, device=dev).view(n, 1) % period
        c = torch.arange(C, dtype=torch.int64, device=dev).view(1, C)
        return (32768.0 + (m * C + c).double() / 64.0).float()

    def decode(v, C, period):
        k = ((v.double() - 32768.0) * 64.0).round().long()
        return (k // C) % period, k % C

    # ------------------------------------------------------------------------------------------------------------------ cases
    def chain_cases(B):
        n = B * H * W
        out = []
        for want_z in (True, False):
            for proj in (False, True):
                w2, w3 = rnd(64, 64, 3, 3, scale=(64 * 9) ** -0.5), rnd(256, 64, 1, 1, scale=64 ** -0.5)
                w1 = rnd(64, 256, 1, 1, scale=256 ** -0.5) if want_z else None
                wds = rnd(256, 64, 1, 1, scale=64 ** -0.5) if proj else None
                b2, b3, b1 = rnd(64, scale=0.3).to(dev), rnd(256, scale=0.3).to(dev), (rnd(64, scale=0.3).to(dev) if want_z else None)
                w2p, s2 = ops.conv_pack_weights_kxr(w2.to(dev), geo)
                tail, s3, s1 = ops.chain_pack_tail(w3.to(dev), w1.to(dev) if want_z else None, wds.to(dev) if proj else None)
                sets = [(ops.split_planes(rnd(n, 64).abs().to(dev), 1), ops.split_planes(rnd(n, 64 if proj else 256).abs().to(dev), 1)) for _ in range(args.sets)]

                def launch(k, outs, w2p=w2p, tail=tail, b2=b2, b3=b3, b1=b1, sc=(s2, s3, s1), sets=sets, want_z=want_z, proj=proj):
                    y, z = outs
                    ops.bottleneck_chain(sets[k][0], sets[k][1], w2p, tail, b2, b3, b1, sc, B, H, W, y=y, z=z, want_z=want_z, proj=proj)

                def fresh(n=n, want_z=want_z):
                    return (torch.empty(2, 8, n, 32, device=dev, dtype=torch.float16), torch.empty(2, 2, n, 32, device=dev, dtype=torch.float16) if want_z else None)

                out.append((f"conv_chain_kernel<z={int(want_z)},proj={int(proj)}> B={B} random", launch, fresh, None))
        # known-answer sets on the identity-shortcut, z-producing instantiation (the one that failed) and its z-less sibling
        for want_z in (True, False):
            # code-x: mid2 == 0, y == x
            w2, w3 = rnd(64, 64, 3, 3, scale=(64 * 9) ** -0.5), rnd(256, 64, 1, 1, scale=64 ** -0.5)
            w1 = torch.zeros(64, 256, device=dev)
            w1[:, 64:128] = torch.eye(64, device=dev)
            w2p, s2 = ops.conv_pack_weights_kxr(w2.to(dev), geo)
            tail, s3, s1 = ops.chain_pack_tail(w3.to(dev), w1.view(64, 256, 1, 1).to(dev) if want_z else None, None)
            b2 = torch.full((64,), -1.0e4, device=dev)
            period = 4096
            xs = [coded(n, 256, period).roll(977 * k, 0) for k in range(args.sets)]
            sets = [(ops.split_planes(rnd(n, 64).abs().to(dev), 1), ops.split_planes(x.to(dev), 1)) for x in xs]
            exp_sets = [s[1] for s in sets]

            def launch(k, outs, w2p=w2p, tail=tail, b2=b2, sc=(s2, s3, s1), sets=sets, want_z=want_z):
                y, z = outs
                ops.bottleneck_chain(sets[k][0], sets[k][1], w2p, tail, b2, None, None, sc, B, H, W, y=y, z=z, want_z=want_z, proj=False)

            def fresh(n=n, want_z=want_z):
                return (torch.empty(2, 8, n, 32, device=dev, dtype=torch.float16), torch.empty(2, 2, n, 32, device=dev, dtype=torch.float16) if want_z else None)

            out.append((f"conv_chain_kernel<z={int(want_z)},proj=0> B={B} code-x (y == shortcut)", launch, fresh, ("x", exp_sets, 256, period)))
            # code-m: y[c] == mid1[c % 64], z == mid1
            w2 = torch.zeros(64, 64, 3, 3, device=dev)
            w2[:, :, 1, 1] = torch.eye(64, device=dev)
            w3 = torch.cat([torch.eye(64, device=dev)] * 4, 0).view(256, 64, 1, 1)
            w2p, s2 = ops.conv_pack_weights_kxr(w2.to(dev), geo)
            tail, s3, s1 = ops.chain_pack_tail(w3.to(dev), w1.view(64, 256, 1, 1).to(dev) if want_z else None, None)
            period = 16384
            ms = [coded(n, 64, period).roll(1409 * k, 0) for k in range(args.sets)]
            zero_x = ops.split_planes(torch.zeros(n, 256, device=dev), 1)
            sets = [(ops.split_planes(m.to(dev), 1), zero_x) for m in ms]
            exp_sets = [s[0] for s in sets]

            def launch(k, outs, w2p=w2p, tail=tail, sc=(s2, s3, s1), sets=sets, want_z=want_z):
                y, z = outs
                ops.bottleneck_chain(sets[k][0], sets[k][1], w2p, tail, None, None, None, sc, B, H, W, y=y, z=z, want_z=want_z, proj=False)

            out.append((f"conv_chain_kernel<z={int(want_z)},proj=0> B={B} code-m (y == mid1 copies)", launch, fresh, ("m", exp_sets, 64, period)))
        return out

    def conv_cases(B, which):
        out = []
        shapes = []
        if "kxr" in which:
            # head output layers (prediction_head_FC.py:157-195): 41-channel class group, 3x3 and 5x3 (kw = 3) / 3x5 (kw = 5) on the 48x80 level
            shapes += [("conv_kxr_kernel<3,2> 256->41 3x3", 256, 48, 3, 3, 48, 80, None), ("conv_kxr_kernel<5,2> 256->41 3x5", 256, 48, 3, 5, 48, 80, None),
                       ("conv_kxr_kernel<3,2> 256->41 5x3", 256, 48, 5, 3, 48, 80, None)]
        if "planar" in which:
            # (the 3x3 tower runs on conv_planar_kx3_kernel -- kx-reuse staging, round 4 -- where its grid takes 256-pixel tiles, i.e. at 32 clips, and on
            # the 128 x 128 ring tiles of conv_planar_kernel at 4; the 1x1 DCN GEMM keeps conv_planar_kernel's 256 x 128 ring at 32 clips)
            shapes += [("conv_planar_kx3_kernel / conv_planar_kernel ring 256->256 3x3 (head tower)", 256, 256, 3, 3, 48, 80, 128),
                       ("conv_planar_kernel ring 256x128 1152->128 1x1 (DCN GEMM)", 1152, 128, 1, 1, 48, 80, 128),
                       ("conv_planar_kernel ring 128x64 256->64 1x1 (bottleneck conv1, two-buffer/ring rule)", 256, 64, 1, 1, 96, 160, 64),
                       ("conv_planar_kernel 128x64 64->256 1x1 + residual (bottleneck conv3)", 64, 256, 1, 1, 96, 160, 64)]
        for name, C, O, kh, kw, h, w, tile in shapes:
            n = B * h * w
            wt, bs = rnd(O, C, kh, kw, scale=(C * kh * kw) ** -0.5).to(dev), rnd(O, scale=0.3).to(dev)
            conv = PlanarConv(wt, bs, 1, (kh // 2, kw // 2), relu=True, fmt=1, tile_n=tile, group_cout=[41] if "kxr" in name else None)
            if "kxr" in name:
                conv.kxr_min_pixels = 0
                if not conv.kxr:
                    raise SystemExit(f"ring_stress: {name} does not route to conv_kxr")
            with_res = "residual" in name
            sets = [(ops.split_planes(rnd(n, C).abs().to(dev), 1), ops.split_planes(rnd(n, O).abs().to(dev), 1) if with_res else None) for _ in range(args.sets)]

            def launch(k, outs, conv=conv, sets=sets, B=B, h=h, w=w):
                conv(sets[k][0], ("img", B, h, w), out_planes=outs[0], residual=sets[k][1])

            def fresh(n=n, O=O):
                return (torch.empty(2, -(-O // 32), n, 32, device=dev, dtype=torch.float16), None)

            out.append((f"{name} B={B}", launch, fresh, None))
        return out

    def window_cases(B):
        """TemporalNet's border-class window set (stm_conv2d_planar_windows_f32): the kw = 3 classes on conv_planar_kx3_kernel<WIN>, the others on
        conv_planar_kernel<CLS>, one 3x3 layer 512 -> 512 on 7x7 RoI maps, 112 RoIs per clip."""
        from stmask_amd.planar import border_windows, _BORDER_CLASSES
        n, C, O, h, w = 112 * B, 512, 512, 7, 7
        wt, bs = rnd(O, C, 3, 3, scale=(C * 9) ** -0.5).to(dev), rnd(O, scale=0.3).to(dev)
        wsc = ops._pow2_wscale(wt)
        wins, packed = [], []
        for ci, win in border_windows(h, w):
            _, _, k0y, k1y, k0x, k1x = _BORDER_CLASSES[ci]
            wins.append(win)
            packed.append(ops.conv_pack_weights(wt[:, :, k0y:k1y, k0x:k1x].contiguous(), tile_n=128, fmt=1, wscale=wsc)[0])
        sets = [ops.split_planes(rnd(n * h * w, C).abs().to(dev), 1) for _ in range(args.sets)]

        def launch(k, outs):
            ops.conv2d_planar_windows(sets[k], packed, wins, bs, n, h, w, C, O, h, w, 1.0 / wsc, relu=True, out_planes=outs[0])

        def fresh():
            return (torch.empty(2, O // 32, n * h * w, 32, device=dev, dtype=torch.float16), None)

        return [(f"window set 3x3 512->512 on 7x7 maps (conv_planar_kx3_kernel<WIN> + conv_planar_kernel<CLS>) B={B}", launch, fresh, None)]

    def dcn_cases(B):
        """dcn_fused_kernel (round 5): weight ring by LDS-DMA + operand ring written by producer waves, read by consumer waves across counted barriers --
        the 128-channel stride-1 layer and the 256-channel stride-2 layer (two channel tiles per patch) of the R50 backbone, bench-like offsets."""
        out = []
        for name, C, H, W, st in (("dcn_fused_kernel 128ch 48x80 stride 1", 128, 48, 80, 1), ("dcn_fused_kernel 256ch 48x80 stride 2", 256, 48, 80, 2)):
            Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
            M = B * Ho * Wo
            wt, bs = rnd(C, C, 3, 3, scale=(9 * C) ** -0.5).to(dev), rnd(C, scale=0.3).to(dev)
            conv = PlanarConv(wt, bs, st, 1, relu=True, fmt=1)
            om = torch.cat([torch.rand(1, 18, device=dev) * 4 - 2 + 0.05 * rnd(M, 18), rnd(M, 9), torch.zeros(M, 5, device=dev)], 1).contiguous()
            sets = [rnd(B * H * W, C).abs().to(dev) for _ in range(args.sets)]

            def launch(k, outs, conv=conv, sets=sets, B=B, H=H, W=W, om=om, st=st):
                conv.deform(sets[k], B, H, W, om, st, 1, 1, has_mask=True, out=outs[0])

            def fresh(M=M, C=C):
                return (torch.empty(2, C // 32, M, 32, device=dev, dtype=torch.float16), None)

            out.append((f"{name} B={B}", launch, fresh, None))
        return out

    which = args.cases.split(",")
    cases = []
    for B in [int(b) for b in args.clips.split(",")]:
        if "chain" in which:
            cases += chain_cases(B)
        cases += conv_cases(B, which)
        if "planar" in which:
            cases += window_cases(B)
        if "dcn" in which:
            cases += dcn_cases(B)

    # ------------------------------------------------------------------------------------------------------------------ solo references
    NAN16 = float("nan")
    prepared = []
    for name, launch, fresh, known in cases:
        refs = []
        for k in range(args.sets):
            outs = fresh()
            for o in outs:
                if o is not None:
                    o.fill_(NAN16)
            launch(k, outs)
            torch.cuda.synchronize()
            refs.append(tuple(o.view(torch.int16).clone() if o is not None else None for o in outs))
            if known is not None:
                kind, exp_sets, C, period = known
                y = refs[k][0]
                if kind == "x":
                    ok = torch.equal(y, exp_sets[k].view(torch.int16))
                else:
                    m = exp_sets[k].view(torch.int16)          # [2, 2, n, 32] -> y slabs 2q, 2q+1 are mid1's slabs 0, 1
                    ok = all(torch.equal(y[:, 2 * q:2 * q + 2], m) for q in range(4)) and (refs[k][1] is None or torch.equal(refs[k][1], m))
                if not ok:
                    print(f"!! {name}: the solo launch of input set {k} does not reproduce its known answer bit for bit", flush=True)
        # a second solo pass must agree with the first (run-to-run determinism without contention)
        for k in range(args.sets):
            outs = fresh()
            for o in outs:
                if o is not None:
                    o.fill_(NAN16)
            launch(k, outs)
            torch.cuda.synchronize()
            for o, r in zip(outs, refs[k]):
                if o is not None and not torch.equal(o.view(torch.int16), r):
                    print(f"!! {name}: two SOLO launches of input set {k} differ", flush=True)
        prepared.append((name, launch, fresh, known, refs))
    torch.cuda.synchronize()
    print(f"ring_stress: {len(prepared)} cases prepared, solo references taken; hammer = {args.hammer}", flush=True)
    if hammer is not None:
        open(hammer[1]["go"], "w").write("go\n")
        time.sleep(1.0)

    # ------------------------------------------------------------------------------------------------------------------ the stress
    def describe(name, o_idx, cur, ref, known, k):
        d = cur != ref
        idx = d.nonzero()
        planes, slabs = idx[:, 0].unique().tolist(), idx[:, 1].unique().tolist()
        px = idx[:, 2].unique()
        tiles = (px // 128).unique()
        msg = (f"    output {o_idx}: {idx.shape[0]} halfwords differ; planes {planes} slabs {slabs} pixels {px.numel()} in tiles {tiles.tolist()[:10]}"
               f"{' ...' if tiles.numel() > 10 else ''}; pixel offsets in tile {sorted((px % 128).unique().tolist())[:40]}; lanes (chunk) {sorted((idx[:, 3] // 8).unique().tolist())}")
        if known is not None and o_idx == 0:
            kind, exp_sets, C, period = known
            from stmask_amd.ops import planes_to_f32
            p0 = px[:64]
            got = planes_to_f32(cur.view(torch.float16)[:, :, p0].contiguous())        # [64 px, 256]
            want = planes_to_f32(ref.view(torch.float16)[:, :, p0].contiguous())
            bad = (got != want).nonzero()[:12]
            for (i, c) in bad.tolist():
                gv, wv = float(got[i, c]), float(want[i, c])
                gm, gc = decode(got[i, c], C, period)
                wm, wc = decode(want[i, c], C, period)
                msg += (f"\n      pixel {int(p0[i])} (tile {int(p0[i]) // 128}, row {int(p0[i]) % 128}) channel {c}: got {gv!r} = code(pixel%{period}={int(gm)}, ch={int(gc)}), "
                        f"expected {wv!r} = code({int(wm)}, {int(wc)}); pixel delta {int(gm) - int(wm)}")
        return msg

    total_bad = 0
    for name, launch, fresh, known, refs in prepared:
        bufs = [fresh() for _ in range(args.batch)]
        n_bad, first_msgs, t0 = 0, [], time.time()
        moved = stalled = 0
        done = 0
        while done < args.launches:
            ks = []
            for b in range(args.batch):
                k = (done + b) % args.sets
                ks.append(k)
                for o in bufs[b]:
                    if o is not None:
                        o.fill_(NAN16)
                if dbg is not None:
                    dbg.zero_()
                launch(k, bufs[b])
                if dbg is not None:
                    torch.cuda.synchronize()
                    rec = dbg.view(-1, 4).cpu()
                    used = rec[:, 3] > 0
                    id0, id1 = rec[used, 0] & 0xFFFFFFFF, (rec[used, 0] >> 32) & 0xFFFFFFFF
                    mv = int((id0 != id1).sum())
                    st = int((rec[used, 1] > 20000).sum())         # s_memrealtime ticks of 10 ns: > 200 us between two barriers
                    moved += mv
                    stalled += st
                    if (mv or st) and len(first_msgs) < 6:
                        w = (rec[used, 1]).argmax()
                        first_msgs.append(f"    launch {done + b}: {mv} waves changed HW_ID, {st} waves waited > 200 us between two barriers (max {int(rec[used, 1].max()) / 100:.0f} us at stage {int(rec[used, 2][w])})")
            torch.cuda.synchronize()
            for b, k in enumerate(ks):
                for o_idx, (o, r) in enumerate(zip(bufs[b], refs[k])):
                    if o is None:
                        continue
                    cur = o.view(torch.int16)
                    if not torch.equal(cur, r):
                        n_bad += 1
                        if len(first_msgs) < 6:
                            first_msgs.append(f"  launch {done + b} (input set {k}):\n" + describe(name, o_idx, cur, r, known, k))
            done += args.batch
        dt = time.time() - t0
        status = "ok" if n_bad == 0 else f"{n_bad} OUTPUTS DIFFER"
        extra = f"; probe: {moved} waves moved, {stalled} stalled > 200 us" if dbg is not None else ""
        print(f"{name:100s} {done:5d} launches beside '{args.hammer}': {status}{extra} ({dt:.1f} s)", flush=True)
        for m in first_msgs:
            print(m, flush=True)
        results.append({"case": name, "launches": done, "hammer": args.hammer, "differing_outputs": n_bad, "waves_moved": moved if dbg is not None else None,
                        "waves_stalled": stalled if dbg is not None else None, "seconds": round(dt, 1)})
        total_bad += n_bad
        del bufs
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"hammer": args.hammer, "library": _lib.LIB_PATH, "results": results}, f, indent=1)
    print(f"ring_stress: {sum(r['launches'] for r in results)} launches, {total_bad} differing outputs", flush=True)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
