"""Loop calls from scored pixels: the clustering behind `peakachu pool`.

Mirror of the reference's peakachu/peakacluster.py (same function names, arguments and
return values), written from its behaviour, not from its text:

* `parse_peakachu`    peakachu/peakacluster.py:7-33
* `second_run`        peakachu/peakacluster.py:35-54
* `find_anchors`      peakachu/peakacluster.py:56-95
* `local_clustering`  peakachu/peakacluster.py:97-172 (with its helper `_cluster_core`)

This stage is serial host work on a few thousand pixels per chromosome (the reference runs
it on the CPU as well); nothing here touches the GPU.  Third-party pieces the reference
calls are called here too (`scipy.signal.find_peaks` / `peak_widths`); its DBSCAN calls all
use `min_samples=2`, for which DBSCAN is exactly "connected components of the graph that
joins points at Euclidean distance <= eps, isolated points are noise" -- implemented
directly (`_dbscan2`), labels numbered like sklearn's (by first member).

Quirks kept on purpose (each is visible in the output):
* a chromosome with fewer than two cluster representatives yields no loops at all
  (peakachu/peakacluster.py:27-30);
* while a cluster grows, its seed pixel is counted twice in the centroid (it is in the
  member list from the start and is appended again on the first pass, :121-131);
* overlapping anchors are merged with the FIRST overlapping record only (:75-84);
* centroids and radii are rounded with numpy's round-half-to-even.
"""
from collections import Counter, defaultdict

import numpy as np
from scipy.signal import find_peaks, peak_widths


def _dbscan2(points, eps):
    """sklearn.cluster.dbscan(points, eps=eps, min_samples=2)[1] for integer 2-D points."""
    pts = np.asarray(points, dtype=np.int64).reshape(-1, 2)
    n = pts.shape[0]
    cell = max(1, int(np.ceil(eps)))
    buckets = defaultdict(list)
    for idx in range(n):
        buckets[(int(pts[idx, 0]) // cell, int(pts[idx, 1]) // cell)].append(idx)
    lim = float(eps) * float(eps)
    labels = np.full(n, -1, dtype=np.int64)
    nxt = 0
    for seed in range(n):
        if labels[seed] != -1:
            continue
        stack, members = [seed], []
        seen = {seed}
        while stack:
            u = stack.pop()
            members.append(u)
            bx, by = int(pts[u, 0]) // cell, int(pts[u, 1]) // cell
            for cx in (bx - 1, bx, bx + 1):
                for cy in (by - 1, by, by + 1):
                    for v in buckets.get((cx, cy), ()):
                        if v in seen:
                            continue
                        dx = int(pts[u, 0]) - int(pts[v, 0])
                        dy = int(pts[u, 1]) - int(pts[v, 1])
                        if dx * dx + dy * dy <= lim:
                            seen.add(v)
                            stack.append(v)
        if len(members) > 1:
            labels[members] = nxt
            nxt += 1
    return labels


def _dist(a, b):
    return float(np.sqrt(float((int(a[0]) - int(b[0])) ** 2 + (int(a[1]) - int(b[1])) ** 2)))


def parse_peakachu(fil, thre, res):
    """Read a score_chromosome / score_genome bedpe, keep pixels with prob >= thre and
    cluster every chromosome.  Returns ({chrom: [(i, j), ...]}, {chrom: {(i, j): [prob, signal]}})."""
    signal_of = defaultdict(dict)
    score_pool = defaultdict(dict)
    with open(fil, "r") as src:
        for line in src:
            col = line.rstrip().split()
            prob = float(col[6])
            if prob >= thre:
                key = (int(col[1]) // res, int(col[4]) // res)
                signal_of[col[0]][key] = float(col[7])
                score_pool[col[0]][key] = [prob, float(col[7])]
    loops = {}
    for chrom, pixels in signal_of.items():
        ranked = [[pixels[rep], rep] for rep, _, _ in local_clustering(pixels, min_count=3, r=2)
                  if rep in pixels]
        ranked.sort(reverse=True)
        loops[chrom] = second_run(ranked) if len(ranked) > 1 else []
    return loops, score_pool


def second_run(sort_list):
    """Representatives closer than 3 bins are one loop: the strongest of each group stays."""
    coords = np.r_[[entry[1] for entry in sort_list]]
    labels = _dbscan2(coords, 3)
    taken, kept = set(), []
    for k, (_, pix) in enumerate(sort_list):
        if pix in taken:
            continue
        kept.append(pix)
        if labels[k] == -1:
            taken.add(pix)
        else:
            for m in coords[labels == labels[k]]:
                taken.add((int(m[0]), int(m[1])))
    return kept


def find_anchors(pos, min_count=3, min_dis=2, wlen=4):
    """Bins that many scored pixels share: peaks of the per-bin pixel count, each with the
    interval its base spans; returns a set of (summit, left, right)."""
    freq = Counter(pos)
    first = min(freq)
    track = np.r_[[freq[b] for b in range(first, max(freq) + 1)]]
    tops = find_peaks(track, height=min_count, distance=min_dis)[0]
    anchors = set()
    owner = {}
    for _, k in sorted(((track[k], k) for k in tops), reverse=True):  # tallest first
        left, right = peak_widths(track, [k], rel_height=1, wlen=wlen)[2:4]
        lo = first + int(np.round(left[0]))
        hi = first + int(np.round(right[0]))
        summit = first + int(k)
        if anchors:
            hit = next((owner[b] for b in range(lo, hi + 1) if b in owner), None)
            if hit is not None:  # grow the first record this interval touches
                anchors.remove(hit)
                summit, lo, hi = hit[0], min(lo, hit[1]), max(hi, hit[2])
        rec = (summit, lo, hi)
        anchors.add(rec)
        for b in range(lo, hi + 1):
            owner[b] = rec
    return anchors


def _cluster_core(sort_list, r, visited, final_list):
    """Greedy clustering of value-sorted pixels: every DBSCAN(eps=r) group is consumed
    strongest pixel first; a cluster absorbs the group members within its radius, moves
    its centre to their (rounded) mean and widens, until nothing more falls in."""
    coords = np.r_[[entry[1] for entry in sort_list]]
    if len(coords) < 2:
        return
    labels = _dbscan2(coords, r)
    used = set()
    for k, (_, seed) in enumerate(sort_list):
        if seed in used or labels[k] == -1:
            continue
        waiting = [tuple(q) for q in coords[labels == labels[k]]]
        centre, radius = seed, r
        members = [seed]
        left_last = -1
        while len(waiting):
            outside = []
            for q in waiting:
                if q in used:
                    continue
                if _dist(q, centre) <= radius:
                    members.append(q)
                else:
                    outside.append(q)
            if len(outside) == left_last:
                break
            left_last = len(outside)
            centre = tuple(np.r_[members].mean(axis=0).round().astype(int))
            radius = np.int64(np.round(max(_dist(centre, q) for q in members))) + r
            waiting = outside
        used.update(members)
        final_list.append((seed, centre, radius))
    visited.update(used)


def local_clustering(Donuts, min_count=3, r=2):
    """Cluster the pixels of one chromosome ({(i, j): signal}); returns a list of
    (representative pixel, centre, radius)."""
    final_list = []
    xs = np.r_[[k[0] for k in Donuts]]
    ys = np.r_[[k[1] for k in Donuts]]
    if xs.size == 0:
        return final_list
    x_anchors = find_anchors(xs, min_count=min_count, min_dis=r)
    y_anchors = find_anchors(ys, min_count=min_count, min_dis=r)
    visited = set()
    present = set(zip(xs, ys))

    def ranked(pixels):
        out = [(Donuts[p], p) for p in pixels]
        out.sort(reverse=True)
        return out

    # pixels inside an (x anchor) x (y anchor) box are clustered box by box ...
    for _, xlo, xhi in x_anchors:
        for _, ylo, yhi in y_anchors:
            box = [(i, j) for i in range(xlo, xhi + 1) for j in range(ylo, yhi + 1)
                   if (i, j) in present]
            _cluster_core(ranked(box), r, visited, final_list)
    # ... then everything no box cluster has absorbed ...
    rest = [(i, j) for i, j in zip(xs, ys) if (i, j) not in visited]
    _cluster_core(ranked(rest), r, visited, final_list)
    # ... and a lone pixel still counts if it sits on an anchor summit
    x_tops = {a[0] for a in x_anchors}
    y_tops = {a[0] for a in y_anchors}
    for i, j in zip(xs, ys):
        if (i, j) not in visited and (i in x_tops or j in y_tops):
            final_list.append(((i, j), (i, j), 0))
    return final_list
