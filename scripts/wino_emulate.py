"""CPU emulation of conv3x3_wino_kernel's data flow (csrc/conv3x3_wino.hip): the packed-weight layout, the LDS images the DMAs build,
every lane's fragment addresses, the MFMA operand / result layout, the two-group output transform and the epilogue's pixel mapping --
index formula by index formula -- against torch's conv2d.  A developer check of the kernel's bookkeeping that needs no GPU
(`python scripts/wino_emulate.py`); the GPU parity tests are tests/test_engine.py::test_winograd_*."""
import numpy as np
import torch
import torch.nn.functional as F

OOB = None


def pack(w):  # wino_pack_kernel
    cout, cin = w.shape[:2]
    n_cs, n_cb = cin // 16, cout // 64
    out = np.zeros(16 * cin * cout, np.float32)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    for o in range(cout):
        for c in range(cin):
            U = G @ w[o, c].astype(np.float64) @ G.T
            cs, h8, hi, c4, cb, col = c >> 4, (c >> 3) & 1, (c >> 2) & 1, c & 3, o >> 6, o & 63
            for i in range(4):
                for j in range(4):
                    block = (((i * 4 + j) * n_cs + cs) * 2 + h8) * n_cb + cb
                    out[block * 512 + (hi * 64 + col) * 4 + c4] = np.float32(U[i, j])
    return out


def plan_windows(nb, ho, wo):  # wino_plan
    if ho <= 8 and wo <= 8:
        return None
    ty_, tx_ = (ho + 1) // 2, (wo + 1) // 2
    busy16 = ho * wo / (((ho + 15) // 16) * ((wo + 15) // 16) * 256)
    best, best_busy, bu = None, busy16, 0
    if busy16 >= 0.9:
        return None
    for wty in range(1, 9):
        for wtx in range(1, 17):
            wt = wty * wtx
            if wt > 64:
                continue
            wrow = (wtx + 1) * 9
            wimg = ((2 * wty + 2) * wrow + 15) // 16 * 16
            wx, wy = -(-tx_ // wtx), -(-ty_ // wty)
            for wg in range(64 // wt, max(0, 64 // wt - 3), -1):
                if wg * wimg > 2560 or wg > 16:
                    continue
                blocks = -(-nb * wx * wy // wg)
                busy = nb * ho * wo / (blocks * 256)
                units = wg * wimg
                if busy > best_busy + 0.02 or (best is not None and busy > best_busy - 1e-9 and units < bu):
                    best, best_busy, bu = dict(wg=wg, wty=wty, wtx=wtx, wrow=wrow, wimg=wimg, wins_x=wx, wins_y=wy), busy, units
    return best


def run_block(x, upk, geo, mt_id, cb, pad, ho, wo, tiles_x, tiles_per_image):
    n, h, w, cin = x.shape
    cout = upk.size // (16 * cin)
    RT = isinstance(geo, dict)
    if RT:
        G, TH, TW, PH, PWD, ROW, IMG = 0, 16, 16, 0, 0, geo["wrow"], 2560
        n_windows = n * geo["wins_x"] * geo["wins_y"]
    else:
        G, TH, TW, PH, PWD, ROW, IMG = geo
    NT = 512

    def px_unit(px):
        return (px >> 1) * 9 + (px & 1) * 4
    A_UNITS = ((1 if RT else G) * IMG + 63) // 64 * 64
    NA = (A_UNITS + NT - 1) // NT
    n_cs, n_cb = cin // 16, cout // 64
    pos_stride = n_cs * n_cb * 4096
    img = 0 if RT else (mt_id // tiles_per_image if G == 1 else mt_id * G)
    trem = mt_id - img * tiles_per_image if G == 1 else 0
    ty0, tx0 = (trem // tiles_x) * TH, (trem % tiles_x) * TW
    xf = x.reshape(-1)
    acc = np.zeros((8, 4, 2, 32, 32), np.float32)  # [wave][j][channel tile][row][col]
    for cs in range(n_cs):
        # ---- patch image
        abuf = np.zeros((A_UNITS, 4), np.float32)
        for r in range(NA):
            for tid in range(NT):
                un = NT * r + tid
                wave = tid >> 6
                if NT * r + wave * 64 >= A_UNITS:
                    continue  # dump
                if RT:
                    if NT * r >= geo["wg"] * geo["wimg"]:
                        continue
                    g, ug = divmod(un, geo["wimg"])
                    py, rem = divmod(ug, geo["wrow"])
                    pair, r9 = divmod(rem, 9)
                    px, chunk = 2 * pair + (r9 >> 2), (4 if r9 == 8 else r9 & 3)
                    win = mt_id * geo["wg"] + g
                    wi, wr = divmod(win, geo["wins_x"] * geo["wins_y"])
                    wy, wx = divmod(wr, geo["wins_x"])
                    iy, ix = wy * 2 * geo["wty"] - pad + py, wx * 2 * geo["wtx"] - pad + px
                    inside = (g < geo["wg"] and win < n_windows and py < 2 * geo["wty"] + 2 and px < 2 * geo["wtx"] + 2 and chunk < 4
                              and 0 <= iy < h and 0 <= ix < w)
                    if inside:
                        off = (((wi * h + iy) * w + ix) * cin * 4 + 16 * chunk + cs * 64) // 4
                        abuf[un] = xf[off:off + 4]
                    continue
                g, ug = divmod(un, IMG)
                py, rem = divmod(ug, ROW)
                pair, r9 = divmod(rem, 9)
                px, chunk = 2 * pair + (r9 >> 2), (4 if r9 == 8 else r9 & 3)
                iy, ix = ty0 - pad + py, tx0 - pad + px
                inside = g < G and img + g < n and py < PH and px < PWD and chunk < 4 and 0 <= iy < h and 0 <= ix < w
                if inside:
                    off = ((((img + g) * h + iy) * w + ix) * cin * 4 + 16 * chunk + cs * 64) // 4
                    abuf[un] = xf[off:off + 4]
        for h8 in range(2):
            s_step = 2 * cs + h8
            # ---- weight stage
            wst = np.zeros((2048, 4), np.float32)
            for q in range(4):
                soff = 4 * q * pos_stride + (s_step * n_cb + cb) * 2048
                for wave in range(8):
                    for lane in range(64):
                        voff = (wave & 1) * 1024 + lane * 16 + (wave >> 1) * pos_stride
                        src = (soff + voff) // 4
                        wst[q * 512 + wave * 64 + lane] = upk[src:src + 4]
            # ---- compute
            for wave in range(8):
                i, wm = wave >> 1, wave & 1
                ra = 0 if i == 0 else (2 if i == 2 else 1)
                rb = 2 if i == 0 else (2 if i == 1 else (1 if i == 2 else 3))
                sg = 1.0 if i == 1 else -1.0
                V = np.zeros((4, 64, 4), np.float32)
                Wv = np.zeros((4, 2, 64, 4), np.float32)
                for lane in range(64):
                    hi = lane >> 5
                    t = 32 * wm + (lane & 31)
                    if RT:
                        wt = geo["wty"] * geo["wtx"]
                        tv = t if t < geo["wg"] * wt else 0
                        g_, rr = divmod(tv, wt)
                        tyy, txx = divmod(rr, geo["wtx"])
                        fa = g_ * geo["wimg"] + 2 * tyy * geo["wrow"] + txx * 9 + hi
                    elif G == 1:
                        fa = 2 * (t >> 3) * ROW + (t & 7) * 9 + hi
                    else:
                        fa = (t >> 4) * IMG + 2 * ((t >> 2) & 3) * ROW + (t & 3) * 9 + hi
                    fa += 2 * h8
                    fb = i * 4 * 128 + hi * 64 + (lane & 31)
                    R = np.zeros((4, 4), np.float32)
                    for c in range(4):
                        R[c] = (abuf[fa + rb * ROW + px_unit(c)] * np.float32(sg) + abuf[fa + ra * ROW + px_unit(c)]).astype(np.float32)
                    V[0, lane] = R[0] - R[2]
                    V[1, lane] = R[1] + R[2]
                    V[2, lane] = R[2] - R[1]
                    V[3, lane] = R[1] - R[3]
                    for j in range(4):
                        for ct in range(2):
                            Wv[j, ct, lane] = wst[fb + j * 128 + ct * 32]
                for j in range(4):
                    for ct in range(2):
                        for k in range(4):
                            A = np.stack([V[j, :32, k], V[j, 32:, k]], axis=1)          # [row][kidx]
                            B = np.stack([Wv[j, ct, :32, k], Wv[j, ct, 32:, k]], axis=0)  # [kidx][col]
                            acc[wave, j, ct] += (A @ B).astype(np.float32)
    # ---- output transform + tiles (two rounds)
    tiles = np.zeros((2, 256, 64), np.float32)

    def to_tile(wave, a, add, neg):
        i, wm = wave >> 1, wave & 1
        m = acc[wave]
        z = [[m[0, ct] + m[1, ct] + m[2, ct] for ct in range(2)], [m[1, ct] - m[2, ct] - m[3, ct] for ct in range(2)]]
        for row in range(32):
            tt = 32 * wm + row
            if RT:
                m00, astep = 4 * tt, 2
            else:
                m00 = 2 * (tt >> 3) * TW + 2 * (tt & 7) if G == 1 else (tt >> 4) * TH * TW + 2 * ((tt >> 2) & 3) * TW + 2 * (tt & 3)
                astep = TW
            for bb in range(2):
                for ct in range(2):
                    v = -z[bb][ct][row] if neg else z[bb][ct][row]
                    dst = tiles[i >> 1, m00 + a * astep + bb, ct * 32:ct * 32 + 32]
                    dst[:] = dst + v if add else v

    for wave in range(8):
        to_tile(wave, (wave >> 1) & 1, False, (wave >> 1) == 3)
    for wave in range(8):
        if wave >> 1 == 1:
            to_tile(wave, 0, True, False)
        if wave >> 1 == 2:
            to_tile(wave, 1, True, True)
    tile = tiles[0] + tiles[1]
    out = {}
    for row in range(256):
        if RT:
            tt, wt = row >> 2, geo["wty"] * geo["wtx"]
            g_, rr = divmod(tt, wt)
            tyy, txx = divmod(rr, geo["wtx"])
            win = mt_id * geo["wg"] + g_
            wi, wr = divmod(win, geo["wins_x"] * geo["wins_y"])
            wy, wx = divmod(wr, geo["wins_x"])
            oy, ox = 2 * (wy * geo["wty"] + tyy) + ((row >> 1) & 1), 2 * (wx * geo["wtx"] + txx) + (row & 1)
            if tt < geo["wg"] * wt and win < n_windows and oy < ho and ox < wo:
                out[(wi, oy, ox)] = tile[row]
            continue
        g, rg = divmod(row, TH * TW)
        oy, ox = ty0 + rg // TW, tx0 + rg % TW
        if oy < ho and ox < wo and img + g < n:
            out[(img + g, oy, ox)] = tile[row]
    return out


def check(n, hw, cin, cout, pad, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, hw, hw, cin)).astype(np.float32)
    wgt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    ho = wo = hw + 2 * pad - 2
    ref = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(wgt), padding=pad).permute(0, 2, 3, 1).numpy()
    upk = pack(wgt)
    small = ho <= 8 and wo <= 8
    win = plan_windows(n, ho, wo)
    geo = win if win is not None else ((4, 8, 8, 10, 10, 50, 512) if small else (1, 16, 16, 18, 18, 84, 18 * 84))
    tiles_y = 1 if small else (ho + 15) // 16
    tiles_x = 1 if small else (wo + 15) // 16
    tiles = (n + 3) // 4 if small else n * tiles_y * tiles_x
    if win is not None:
        tiles = -(-n * win["wins_x"] * win["wins_y"] // win["wg"])
        print("   windows:", win)
    got = np.full_like(ref, np.nan)
    for mt in range(tiles):
        for cb in range(cout // 64):
            for (b, oy, ox), v in run_block(x, upk, geo, mt, cb, pad, ho, wo, tiles_x, tiles_y * tiles_x).items():
                got[b, oy, ox, cb * 64:cb * 64 + 64] = v
    assert not np.isnan(got).any(), "outputs not covered"
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print(f"n={n} hw={hw} cin={cin} cout={cout} pad={pad}: rel err {err:.2e}")
    assert err < 2e-5, err


def bank_check():
    """Every service group of a ds_read_b128 (16 lanes) must hit 16 different 16-byte bank groups, for both geometries."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for G, ROW, IMG in ((1, 84, 18 * 84), (4, 50, 512)):
        for wm in range(2):
            for grp in groups:
                units = []
                for lane in grp:
                    t = 32 * wm + lane
                    fa = 2 * (t >> 3) * ROW + (t & 7) * 9 if G == 1 else (t >> 4) * IMG + 2 * ((t >> 2) & 3) * ROW + (t & 3) * 9
                    units.append(fa % 16)
                assert len(set(units)) == 16, (G, wm, grp, units)
    print("patch reads: conflict-free")


if __name__ == "__main__":
    bank_check()
    if "--windows-only" not in __import__("sys").argv:
        check(1, 16, 16, 64, 1)
        check(5, 7, 32, 64, 1)       # W8 geometry, odd map, a partial block of images
    check(8, 14, 16, 64, 1)      # window geometry: eight windows of 1 x 7 tiles, blocks across images, a partial last block
    check(3, 28, 16, 64, 1)      # window geometry: 3 windows of 3 x 7 tiles
    if "--all" in __import__("sys").argv:
        check(1, 20, 16, 128, 1)     # partial blocks, two channel blocks
        check(2, 12, 16, 64, 0)      # valid convolution
    print("emulation ok")
