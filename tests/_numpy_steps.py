"""Stand-in *steps* providers for stainx_amd.distributed built on numpy and the oracle (TEST CODE).

They implement the same staged interface as the HIP backend classes so that the collective
choreography of ``stainx_amd/distributed.py`` can run on CPU-only machines under gloo."""
from __future__ import annotations

import numpy as np
import torch

from oracle import stain_oracle as so


def float_key(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return np.where(u >> 31 != 0, ~u, u | np.uint32(0x80000000)).astype(np.uint32)


def key_float(k: int) -> np.float32:
    k = np.uint32(k)
    u = (k & np.uint32(0x7FFFFFFF)) if (k >> 31) else ~k
    return np.array([u], dtype=np.uint32).view(np.float32)[0]


def nearest_rank_index(q: float, n: int) -> int:
    return 0 if n == 0 else round(0.01 * q * (n - 1))


class NumpyMacenkoSteps:
    def _od_rows(self, images: torch.Tensor) -> np.ndarray:
        od = so.optical_density(so.to_unit_float(images.numpy()))
        return np.transpose(od, (1, 0, 2, 3)).reshape(3, -1).T          # (n,3)

    def dfit_moments(self, images):
        rows = self._od_rows(images).astype(np.float64)
        keep = self._od_rows(images).min(axis=1) >= so.BETA

        def raw(x):
            return [x.shape[0], *x.sum(0), (x[:, 0] * x[:, 0]).sum(), (x[:, 0] * x[:, 1]).sum(), (x[:, 0] * x[:, 2]).sum(),
                    (x[:, 1] * x[:, 1]).sum(), (x[:, 1] * x[:, 2]).sum(), (x[:, 2] * x[:, 2]).sum()]

        return torch.tensor(raw(rows[keep]) + raw(rows), dtype=torch.float64)

    def dfit_begin(self, moments):
        a = moments.numpy()
        cnt = a[0]
        mean = a[1:4] / cnt
        sxy = np.array([[a[4], a[5], a[6]], [a[5], a[7], a[8]], [a[6], a[8], a[9]]])
        cov = (sxy - np.outer(a[1:4], mean)) / (cnt - 1)
        _, vecs = np.linalg.eigh(cov)
        vecs = vecs[:, [1, 2]]
        vecs = vecs * np.where(vecs.sum(0) < 0, -1.0, 1.0)[None, :]
        n_sel, n_all = int(cnt), int(a[10])
        return {"vecs": vecs.astype(np.float32), "n_sel": n_sel, "n_all": n_all, "rank": [nearest_rank_index(1, n_sel), nearest_rank_index(99, n_sel), 0, 0],
                "prefix": [0, 0, 0, 0], "mask": [0, 0, 0, 0], "round": [0, 0]}

    def _keys(self, images, state, stage):
        rows = self._od_rows(images)
        if stage == 0:
            kept = rows[rows.min(axis=1) >= so.BETA]
            proj = (kept @ state["vecs"]).astype(np.float32)
            k = float_key(np.arctan2(proj[:, 1], proj[:, 0]).astype(np.float32))
            return k, k
        conc = (state["pinv"] @ rows.T).astype(np.float32)
        return float_key(conc[0]), float_key(conc[1])

    def dfit_histogram(self, images, state, stage):
        shift = 24 - 8 * state["round"][stage]
        out = np.zeros((2, 256), dtype=np.int64)
        for j, k in enumerate(self._keys(images, state, stage)):
            s = stage * 2 + j
            member = ((k ^ np.uint32(state["prefix"][s])) & np.uint32(state["mask"][s])) == 0
            out[j] = np.bincount((k[member] >> np.uint32(shift)) & np.uint32(255), minlength=256)
        return torch.from_numpy(out)

    def dfit_advance(self, state, stage, hist):
        h = hist.numpy()
        shift = 24 - 8 * state["round"][stage]
        for j in range(2):
            s = stage * 2 + j
            cum = np.cumsum(h[j])
            d = int(np.searchsorted(cum, state["rank"][s], side="right"))
            d = min(d, 255)
            state["rank"][s] -= int(cum[d - 1]) if d > 0 else 0
            state["prefix"][s] |= d << shift
            state["mask"][s] |= 0xFF << shift
        state["round"][stage] += 1
        if state["round"][stage] < 4:
            return
        if stage == 0:
            state["he"] = so.stain_vectors_from_angles(state["vecs"], key_float(state["prefix"][0]), key_float(state["prefix"][1]))
            state["pinv"] = np.linalg.pinv(state["he"].astype(np.float64)).astype(np.float32)
            state["rank"][2] = state["rank"][3] = nearest_rank_index(99, state["n_all"])
        else:
            state["max_c"] = np.array([key_float(state["prefix"][2]), key_float(state["prefix"][3])], dtype=np.float32)

    def dfit_result(self, state):
        return torch.from_numpy(state["he"].copy()), torch.from_numpy(state["max_c"].copy())


def _sample_geometry(n: int, h: int, w: int) -> tuple[int, int]:
    """(stride, count) of the local pixel sample: the smallest power-of-two stride with at most 4096 samples."""
    total, stride = n * h * w, 1
    while (total + stride - 1) // stride > 4096:
        stride *= 2
    return stride, min(4096, (total + stride - 1) // stride)


def _bracket_ranks(m: int, n_total: int, k0: int) -> tuple[int, int]:
    f = k0 / (n_total - 1) if n_total > 1 else 0.0
    r, sd = f * (m - 1), np.sqrt(m * f * (1 - f))
    return int(min(max(np.floor(r - 6 * sd - 3), 0), m - 1)), int(min(max(np.ceil(r + 6 * sd + 3), 0), m - 1))


class NumpyMacenkoBracketSteps(NumpyMacenkoSteps):
    """The bracket form of the staged pooled fit (``pfit_*``): sample brackets, counts + 256-bin histogram of the keys
    inside them, the candidates of the picked bin, exact selection on their union."""
    device = torch.device("cpu")

    def pfit_sample_count(self, n, h, w):
        return _sample_geometry(n, h, w)[1]

    def pfit_stats(self, images):
        rows = self._od_rows(images)
        mom = self.dfit_moments(images)[:10]
        n, _, h, w = images.shape
        stride, count = _sample_geometry(n, h, w)
        sample = np.zeros((3, 4096), dtype=np.float32)
        sample[:, :count] = rows[::stride][:count].T
        return mom, torch.from_numpy(sample)

    def pfit_plane(self, moments, n_all, sample_union, sample_count, shape):
        full = torch.cat([moments.to(torch.float64), torch.tensor([float(n_all)] + [0.0] * 9, dtype=torch.float64)])
        st = self.dfit_begin(full)
        st["sample"] = sample_union.numpy()[:, :sample_count].T.astype(np.float32)          # (m,3)
        kept = st["sample"][st["sample"].min(axis=1) >= so.BETA]
        proj = (kept @ st["vecs"]).astype(np.float32)
        keys = np.sort(float_key(np.arctan2(proj[:, 1], proj[:, 0]).astype(np.float32)))
        st["brackets"] = [tuple(int(keys[i]) for i in _bracket_ranks(len(keys), st["n_sel"], st["rank"][s])) for s in (0, 1)]
        self.state = st

    def _bin(self, keys, lo, hi):
        vlo, vhi = float(key_float(lo)), float(key_float(hi))
        scale = 256.0 / (vhi - vlo) if vhi > vlo else 0.0
        vals = np.array([key_float(k) for k in keys], dtype=np.float64) if len(keys) < 64 else keys_to_float(keys).astype(np.float64)
        return np.clip(((vals - vlo) * scale), 0, 255).astype(np.int64)

    def pfit_pass(self, images, stage, n_all, sample_count):
        st = self.state
        sums = np.zeros(1033, dtype=np.int64)
        st["cand"] = []
        for j, k in enumerate(self._keys(images, st, stage)):
            lo, hi = st["brackets"][j]
            inside = k[(k >= np.uint32(lo)) & (k <= np.uint32(hi))]
            st["cand"].append(inside)
            sums[stage * 512 + j * 256: stage * 512 + (j + 1) * 256] = np.bincount(self._bin(inside, lo, hi), minlength=256)
            sums[1024 + 2 * stage + j] = int((k < np.uint32(lo)).sum())
            sums[1028 + 2 * stage + j] = len(inside)
        return torch.from_numpy(sums)

    def pfit_gather(self, sums_global, stage, n_all, sample_count, shape, share):
        st, g = self.state, sums_global.numpy()
        compact = np.zeros((2, share), dtype=np.uint32)
        counts = np.zeros(2, dtype=np.int32)
        st["want"] = [0, 0]
        for j in range(2):
            want = st["rank"][2 * stage + j] - int(g[1024 + 2 * stage + j])
            if not 0 <= want < int(g[1028 + 2 * stage + j]):      # a bracket missed: reported through the status word, like the native steps
                st["missed"] = True
                want = 0
            cum = np.cumsum(g[stage * 512 + j * 256: stage * 512 + (j + 1) * 256])
            b = int(np.searchsorted(cum, want, side="right"))
            st["want"][j] = want - (int(cum[b - 1]) if b else 0)
            lo, hi = st["brackets"][j]
            mine = st["cand"][j][self._bin(st["cand"][j], lo, hi) == b]
            counts[j] = len(mine)
            compact[j, : len(mine)] = mine
        return torch.from_numpy(compact.view(np.int32)), torch.from_numpy(counts)

    def pfit_finish(self, gathered_compact, gathered_counts, stage, n_all, sample_count, shape):
        st = self.state
        keys = []
        for j in range(2):
            parts = [gathered_compact[r, j, : int(gathered_counts[r, j])].numpy().view(np.uint32) for r in range(gathered_counts.shape[0])]
            merged = np.sort(np.concatenate(parts))
            if st["want"][j] >= len(merged):
                st["missed"] = True
            keys.append(int(merged[min(st["want"][j], len(merged) - 1)]) if len(merged) else 0)
        if stage == 0:
            st["he"] = so.stain_vectors_from_angles(st["vecs"], key_float(keys[0]), key_float(keys[1]))
            st["pinv"] = np.linalg.pinv(st["he"].astype(np.float64)).astype(np.float32)
            st["rank"][2] = st["rank"][3] = nearest_rank_index(99, n_all)
            conc = (st["pinv"] @ st["sample"].T).astype(np.float32)
            st["brackets"] = []
            for j in range(2):
                ks = np.sort(float_key(conc[j]))
                st["brackets"].append(tuple(int(ks[i]) for i in _bracket_ranks(len(ks), n_all, st["rank"][2 + j])))
            return None
        max_c = np.array([key_float(keys[0]), key_float(keys[1])], dtype=np.float32)
        return torch.from_numpy(st["he"].copy()), torch.from_numpy(max_c), torch.full((1,), 1 if st.get("missed") else 0, dtype=torch.int32)


class NumpyMacenkoPackedSteps(NumpyMacenkoBracketSteps):
    """The same steps behind the PACKED interface (include/stainx_hip.h, sx_macenko_pfit_*_packed): one contiguous record per rank and
    exchange, packed and unpacked here the way the library's kernels do it -- so the gloo tests run the choreography the product runs."""
    RECORD = 8 + 80 + 3 * 4096 * 4

    def pfit_stats_packed(self, images):
        mom, sample = self.pfit_stats(images)
        rec = np.zeros(self.RECORD, dtype=np.uint8)
        rec[:8] = np.array([images.shape[0]], dtype=np.int64).view(np.uint8)
        rec[8:88] = np.asarray(mom.numpy(), dtype=np.float64).view(np.uint8)
        rec[88:] = np.ascontiguousarray(sample.numpy(), dtype=np.float32).reshape(-1).view(np.uint8)
        return torch.from_numpy(rec)

    def pfit_empty_record(self):
        return torch.zeros(self.RECORD, dtype=torch.uint8)

    def pfit_plane_packed(self, gathered, sample_counts, expected_tiles, n_all, sample_count, shape):
        got = gathered.numpy()
        world = got.shape[0]
        tiles = np.array([got[r, :8].copy().view(np.int64)[0] for r in range(world)])
        moments = np.zeros(10, dtype=np.float64)
        for r in range(world):                                    # rank order, like the kernel
            moments = moments + got[r, 8:88].copy().view(np.float64)
        union = np.zeros((3, 4096), dtype=np.float32)
        base = 0
        for r in range(world):
            sample = got[r, 88:].copy().view(np.float32).reshape(3, 4096)
            cols = sample[:, : int(sample_counts[r]) : world]
            take = min(cols.shape[1], 4096 - base)
            if take > 0:
                union[:, base: base + take] = cols[:, :take]
            base += cols.shape[1]
        assert min(base, 4096) == sample_count, (base, sample_count)
        self.pfit_plane(torch.from_numpy(moments), n_all, torch.from_numpy(union), sample_count, shape)
        stale = 0 if expected_tiles is None else int((tiles != expected_tiles.numpy()).any())
        return torch.tensor([stale], dtype=torch.int32)

    def pfit_gather_packed(self, sums_global, stage, n_all, sample_count, shape, share, stale):
        compact, counts = self.pfit_gather(sums_global, stage, n_all, sample_count, shape, share)
        flag = torch.zeros(1, dtype=torch.int32) if stale is None else stale.to(torch.int32).reshape(1)
        return torch.cat([counts.to(torch.int32).flatten(), flag, compact.to(torch.int32).flatten()])

    def pfit_finish_packed(self, gathered_rows, stage, n_all, sample_count, shape, share):
        world = gathered_rows.shape[0]
        out = self.pfit_finish(gathered_rows[:, 3:].reshape(world, 2, share), gathered_rows[:, :2], stage, n_all, sample_count, shape)
        if out is None:
            return None
        he, max_c, status = out
        return he, max_c, status + 16 * int(gathered_rows[:, 2].max())


def keys_to_float(keys: np.ndarray) -> np.ndarray:
    k = np.ascontiguousarray(keys, dtype=np.uint32)
    u = np.where(k >> 31 != 0, k & np.uint32(0x7FFFFFFF), ~k)
    return u.astype(np.uint32).view(np.float32)


class NumpyReinhardSteps:
    def local_sums(self, images):
        lab = so.rgb_to_lab(so.to_unit_float(images.numpy())).astype(np.float64) - 128.0
        flat = np.transpose(lab, (1, 0, 2, 3)).reshape(3, -1)
        return torch.from_numpy(np.concatenate([flat.sum(1), (flat * flat).sum(1)]))

    def apply_with_sums(self, images, sums, n_total, ref_mean, ref_std):
        s = sums.numpy()
        mean = s[:3] / n_total
        std = np.sqrt((s[3:] - s[:3] * mean) / (n_total - 1))
        x = images.numpy()
        lab = so.rgb_to_lab(so.to_unit_float(x))
        rm = np.asarray(ref_mean, dtype=np.float32).reshape(1, 3, 1, 1)
        rs = np.asarray(ref_std, dtype=np.float32).reshape(1, 3, 1, 1)
        mu = (mean + 128.0).astype(np.float32).reshape(1, 3, 1, 1)
        sd = std.astype(np.float32).reshape(1, 3, 1, 1)
        lab_n = ((lab - mu) / (sd + np.float32(1e-8))) * rs + rm
        rgb = np.clip(so.lab_to_rgb(lab_n.astype(np.float32)), 0, 1)
        return torch.from_numpy(so.restore_dtype(rgb, x.dtype, in_0_255=False))


class NumpyHMSteps:
    def local_counts(self, images):
        u8, _ = so.images_to_uint8(images.numpy())
        return torch.from_numpy(np.stack([np.bincount(u8[:, c].reshape(-1), minlength=256) for c in range(3)]).astype(np.int64))

    def apply_with_counts(self, images, counts, n_total, ref_hists):
        x = images.numpy()
        u8, scaled = so.images_to_uint8(x)
        out = np.empty(u8.shape, dtype=np.float32)
        for c in range(3):
            lut = so.hm_lut(counts[c].numpy(), np.asarray(ref_hists[c]), n_total)
            out[:, c] = lut[u8[:, c]]
        if scaled:
            return torch.from_numpy(so.restore_dtype(np.clip(out / np.float32(255), 0, 1), x.dtype, in_0_255=False))
        return torch.from_numpy(so.restore_dtype(np.clip(out, 0, 255), x.dtype, in_0_255=True))
