"""Diagnosis helpers on top of m3dreg_debug_cloud_raw (include/m3dreg.h): the search structures of a bucketed cloud read back as they lie in HBM and
checked against the cloud itself. Not used by any product path; tests/test_gpu_pipelined.py and scripts/r6_hunt2.py call it.

The layouts are csrc/m3d_device.h's (M3dTileHdr, M3dTileImgMeta, the tile image: voxel list | staged points | bucket deltas)."""
import numpy as np

TILE_PTS, ECAP, VCAP, PCAP = 512, 512, 1280, 2048
IMG_WORDS = (VCAP * 8 + PCAP * 16 + ECAP * 4) // 4
OVERSIZE = 1


def tile_image_problems(cloud, level=0, max_report=8):
    """Every tile image of a target cloud against the cloud's own sorted points. For every tile that is not flagged oversize:
      * the voxel lists of its images name every voxel once, their {LDS position, population} runs cover the image's staged points exactly once,
      * every staged point IS the sorted point its entry says it is (LDS position + delta[staged bucket] = sorted position): coordinates and input index,
      * the points of one entry lie in one voxel of the cloud (one sorted key), different entries in different voxels.
    Returns a list of strings, empty when the images are sound. Waits for the cloud's bucketing; a few MB of device-to-host copies per cloud."""
    thdr = cloud.raw("thdr", level).reshape(-1, 4)
    timg = cloud.raw("timg", level).reshape(-1, IMG_WORDS)
    meta = cloud.raw("timeta", level).reshape(-1, 2)
    ex = cloud.export(level)
    sxyz = np.ascontiguousarray(ex["sorted_xyz"]).view(np.uint32).reshape(-1, 3)
    skey, perm = ex["sorted_keys"], ex["perm"].astype(np.int64)
    n_valid = int((skey != 0xFFFFFFFF).sum())
    out = []
    for t in range((n_valid + TILE_PTS - 1) // TILE_PTS):
        extra, n_img, flags, meta0 = (int(x) for x in thdr[t])
        if flags & OVERSIZE:
            continue
        n_b = flags >> 16
        if n_b == 0:   # a tile that owns no bucket (its 512 positions lie inside a bucket that starts in an earlier tile): an empty image nobody is sent to
            if not (n_img == 1 and (meta0 & 0xFFFF) == 0 and int(meta[t][0]) == 0 and (int(meta[t][1]) & 0x7FFFFFFF) == 0):
                out.append(f"tile {t}: no staged bucket but header {thdr[t].tolist()}, image meta {meta[t].tolist()}")
            continue
        if not (1 <= n_img <= 32 and 1 <= n_b <= ECAP):
            out.append(f"tile {t}: header {thdr[t].tolist()}")
            continue
        delta = timg[t][2 * VCAP + 4 * PCAP:2 * VCAP + 4 * PCAP + n_b].view(np.int32).astype(np.int64)
        keys_seen = []
        for j in range(n_img):
            image = t if j == 0 else extra + j - 1
            if image >= len(timg):
                out.append(f"tile {t}: image {image} of {len(timg)}")
                break
            npnt, nvx = int(meta[image][0]), int(meta[image][1]) & 0x7FFFFFFF
            if npnt > PCAP or nvx > VCAP or nvx == 0:
                out.append(f"tile {t} image {image}: {npnt} points, {nvx} voxels")
                break
            if j == 0 and npnt != (meta0 & 0xFFFF):
                out.append(f"tile {t}: header says {meta0 & 0xFFFF} staged points, the image's meta {npnt}")
            vl = timg[image][:2 * nvx].reshape(-1, 2)
            pos, cnt, bkt = (vl[:, 1] & 0x7FF).astype(np.int64), (((vl[:, 1] >> 11) & 0x7FF) + 1).astype(np.int64), (vl[:, 1] >> 22).astype(np.int64)
            keys_seen.append(vl[:, 0])
            cover = np.zeros(PCAP + 1, np.int64)
            np.add.at(cover, pos, 1); np.add.at(cover, np.minimum(pos + cnt, PCAP), -1)
            cover = np.cumsum(cover)[:PCAP]
            if (pos + cnt > npnt).any() or (cover[:npnt] != 1).any() or (bkt >= n_b).any():
                out.append(f"tile {t} image {image}: the voxel list does not cover the {npnt} staged points exactly once "
                           f"({int((cover[:npnt] == 0).sum())} uncovered, {int((cover[:npnt] > 1).sum())} covered twice, {int((bkt >= n_b).sum())} bucket numbers out of range)")
                continue
            ent = np.repeat(np.arange(nvx), cnt)                               # entry of every staged point, in entry order
            lds = np.repeat(pos, cnt) + (np.arange(len(ent)) - np.repeat(np.cumsum(cnt) - cnt, cnt))
            spos = lds + delta[bkt[ent]]
            if (spos < 0).any() or (spos >= n_valid).any():
                out.append(f"tile {t} image {image}: {int(((spos < 0) | (spos >= n_valid)).sum())} staged points name sorted positions outside the cloud")
                continue
            P = timg[image][2 * VCAP:2 * VCAP + 4 * npnt].reshape(-1, 4)[lds]
            bad = (P[:, :3] != sxyz[spos]).any(axis=1) | (P[:, 3].astype(np.int64) != perm[spos])
            if bad.any():
                b = np.nonzero(bad)[0]
                out.append(f"tile {t} image {image}: {len(b)} of {npnt} staged points are not the sorted points their entries name (LDS positions {np.unique(lds[b])[:12].tolist()} ...)")
                continue
            k_of = skey[spos]
            first = np.cumsum(cnt) - cnt
            if (k_of != np.repeat(k_of[first], cnt)).any() or len(np.unique(k_of[first])) != nvx:
                out.append(f"tile {t} image {image}: an entry spans several voxels of the cloud, or two entries share one")
        if keys_seen:
            allk = np.concatenate(keys_seen)
            if len(np.unique(allk)) != len(allk):
                out.append(f"tile {t}: {len(allk) - len(np.unique(allk))} voxel keys listed twice")
        if len(out) >= max_report:
            out.append("...")
            break
    return out
