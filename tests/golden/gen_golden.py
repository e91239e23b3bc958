#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by RUNNING the reference's own Python code
in this container (never on the GPU box; /root/reference does not travel).

  - nearest_embed.npz      Graph_Idea/model/model_2_0.py:43-103 (nearest_embed) imported and called
  - mutual_block.npz       the torch calls of Others/SuperPoint_data/models/superglue.py:267-286
                           (einsum / max / gather / where, helper arange_like imported from the file)
                           applied to synthetic unit descriptors; SuperGlue.forward itself cannot run
                           because its weight files are absent from the reference tree
  - matching_ref.txt       get_list_matching_pairs + save_matching_sfm (main.py:114-155), executed
  - pairs_ref.txt          generate_imagePairs (main.py:157-169), executed
  - feature_ref.txt        save_feature_locations (main.py:133-140), executed
  - resort_ref.npz         sort_scores, sort_feature_Or_descriptors, get_correct_new_matching_index
                           (main.py:75-112) executed in the order of the exporter's loop (main.py:253-266)
main.py cannot be imported (it needs cv2 and runs its whole pipeline at import), so its pure helper
functions are compiled from its source text with ast and executed here.

Usage: python tests/golden/gen_golden.py   (writes next to this file)
"""
import ast
import os
import sys

import numpy as np
import pandas as pd
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_main_helpers():
    path = os.path.join(REF, "Others/SuperPoint_data/main.py")
    tree = ast.parse(open(path).read())
    wanted = {"get_list_matching_pairs", "save_matching_sfm", "generate_imagePairs", "save_feature_locations",
              "sort_scores", "sort_feature_Or_descriptors", "get_correct_new_matching_index"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    # module-level constants the helpers' default arguments refer to (main.py:35-36)
    consts = [n for n in tree.body if isinstance(n, ast.Assign) and
              any(isinstance(t, ast.Name) and t.id in ("consecutive", "discrete") for t in n.targets)]
    mod = ast.Module(body=consts + body, type_ignores=[])
    ns = {"np": np, "pd": pd, "os": os}
    exec(compile(mod, path, "exec"), ns)
    return ns


def gen_nearest_embed():
    sys.path.insert(0, os.path.join(REF, "Graph_Idea"))
    from model.model_2_0 import nearest_embed

    g = torch.Generator().manual_seed(1234)
    x = torch.randn((3, 256, 40), generator=g)
    emb = torch.rand((256, 64), generator=g)
    emb[:, 17] = emb[:, 5]     # duplicate codes: the first index must win
    emb[:, 40] = emb[:, 5]
    x[0, :, 3] = emb[:, 5] + 0.01
    x[1, :, 7] = emb[:, 40]
    res, argmin = nearest_embed(x, emb)
    # a second, smaller-dimension case
    x2 = torch.randn((2, 64, 33), generator=g)
    emb2 = torch.rand((64, 512), generator=g)
    res2, argmin2 = nearest_embed(x2, emb2)
    np.savez(os.path.join(OUT, "nearest_embed.npz"),
             x=x.numpy(), emb=emb.numpy(), result=res.numpy(), argmin=argmin.numpy(),
             x2=x2.numpy(), emb2=emb2.numpy(), result2=res2.numpy(), argmin2=argmin2.numpy())


def gen_nearest_embed_near_ties():
    """Near-ties, where a distance computed as |x|^2 + |e|^2 - 2 x.e drowns in rounding and only the
    difference form the reference uses (norm of x - e) ranks correctly: codes in pairs 1e-6 apart, points
    1e-4 .. 1e-2 away from a code of such a pair."""
    sys.path.insert(0, os.path.join(REF, "Graph_Idea"))
    from model.model_2_0 import nearest_embed

    g = torch.Generator().manual_seed(4321)
    k, dim, npts = 128, 256, 96
    emb = torch.rand((dim, k), generator=g)
    for c in range(0, 64, 2):       # 32 pairs of codes 1e-6 apart (per component, random signs)
        emb[:, c + 1] = emb[:, c] + 1e-6 * torch.sign(torch.randn(dim, generator=g))
    x = torch.rand((2, dim, npts), generator=g)
    for b in range(2):
        for p in range(npts):
            c = int(torch.randint(0, 64, (1,), generator=g))
            u = torch.randn(dim, generator=g)
            u /= u.norm()
            x[b, :, p] = emb[:, c] + (10.0 ** float(-4.0 + 2.0 * torch.rand(1, generator=g))) * u
    res, argmin = nearest_embed(x, emb)
    np.savez(os.path.join(OUT, "nearest_embed_near_ties.npz"), x=x.numpy(), emb=emb.numpy(), result=res.numpy(),
             argmin=argmin.numpy())


def gen_mutual_block():
    sys.path.insert(0, os.path.join(REF, "Others/SuperPoint_data"))
    from models.superglue import arange_like

    g = torch.Generator().manual_seed(99)

    def unit(d, n):
        v = torch.randn((1, d, n), generator=g)
        return torch.nn.functional.normalize(v, p=2, dim=1)

    cases = {}
    for tag, (n, m) in {"a": (120, 150), "b": (64, 40)}.items():
        d0, d1 = unit(256, n), unit(256, m)
        k = min(n, m) // 3
        src = torch.randperm(n, generator=g)[:k]
        dst = torch.randperm(m, generator=g)[:k]
        pert = d0[:, :, src] + 0.05 * torch.randn((1, 256, k), generator=g)
        d1[:, :, dst] = torch.nn.functional.normalize(pert, p=2, dim=1)
        match_threshold = 0.2
        # superglue.py:267-268
        scores = torch.einsum('bdn,bdm->bnm', d0, d1)
        scores = scores / 256 ** .5
        # superglue.py:276-286 (on the raw score matrix: no learned layers, no Sinkhorn)
        max0, max1 = scores.max(2), scores.max(1)
        indices0, indices1 = max0.indices, max1.indices
        mutual0 = arange_like(indices0, 1)[None] == indices1.gather(1, indices0)
        mutual1 = arange_like(indices1, 1)[None] == indices0.gather(1, indices1)
        zero = scores.new_tensor(0)
        mscores0 = torch.where(mutual0, max0.values.exp(), zero)
        mscores1 = torch.where(mutual1, mscores0.gather(1, indices1), zero)
        valid0 = mutual0 & (mscores0 > match_threshold)
        valid1 = mutual1 & valid0.gather(1, indices1)
        indices0 = torch.where(valid0, indices0, indices0.new_tensor(-1))
        indices1 = torch.where(valid1, indices1, indices1.new_tensor(-1))
        cases["desc0_" + tag] = d0[0].numpy()
        cases["desc1_" + tag] = d1[0].numpy()
        cases["matches0_" + tag] = indices0[0].numpy()
        cases["matches1_" + tag] = indices1[0].numpy()
        cases["mscores0_" + tag] = mscores0[0].numpy()
        cases["mscores1_" + tag] = mscores1[0].numpy()
        cases["argmax0_" + tag] = max0.indices[0].numpy()
        cases["argmax1_" + tag] = max1.indices[0].numpy()
    np.savez(os.path.join(OUT, "mutual_block.npz"), **cases)


def gen_text_formats():
    h = load_main_helpers()
    rng = np.random.default_rng(5)
    # match txt: three records, one with a match on the last keypoint, one empty
    recs = []
    m_a = -np.ones(12); m_a[[1, 4, 11]] = [7, 0, 3]
    m_b = -np.ones(9); m_b[[0, 2, 3, 7]] = [5, 8, 1, 2]
    m_c = -np.ones(5)
    path = os.path.join(OUT, "matching_ref.txt")
    if os.path.exists(path):
        os.remove(path)
    for name0, name1, m in (("0.jpg", "1.jpg", m_a), ("0.jpg", "2.jpg", m_b), ("1.jpg", "2.jpg", m_c)):
        l0, l1, n = h["get_list_matching_pairs"](m)
        h["save_matching_sfm"](path, l0, l1, name0, name1, n)
        recs.append(m.astype(np.int32))
    np.savez(os.path.join(OUT, "matching_ref_inputs.npz"), m_a=recs[0], m_b=recs[1], m_c=recs[2])
    # pair list over 20 images named as create_vsfm_data names them (main.py:64)
    names = [str(i) + ".jpg" for i in range(20)]
    infor = os.path.join(OUT, "_infor_tmp.txt")
    pd.DataFrame({0: ["x"] * 20, 1: names}).to_csv(infor, header=False, index=False, sep=" ")
    pairs = h["generate_imagePairs"](infor_path=infor)
    os.remove(infor)
    with open(os.path.join(OUT, "pairs_ref.txt"), "w") as f:
        for a, b in pairs:
            f.write("%s %s\n" % (a, b))
    # keypoint txt
    kp = (rng.random((17, 2)) * np.array([640, 480])).astype(np.float32)
    np.save(os.path.join(OUT, "feature_ref_inputs.npy"), kp)
    h["save_feature_locations"](kp, os.path.join(OUT, "feature_ref.txt"))


def gen_resort():
    """The exporter's re-sort of one pair (main.py:253-266): keypoints of both images ordered by descending
    detector score, the dense match array carried along and re-indexed.  Scores hold ties."""
    h = load_main_helpers()
    rng = np.random.default_rng(11)
    n0, n1 = 57, 43
    scores0 = np.round(rng.random(n0), 2).astype(np.float32)     # two decimals: ties
    scores1 = np.round(rng.random(n1), 2).astype(np.float32)
    kpts0 = (rng.random((n0, 2)) * np.array([640, 480])).astype(np.float32)
    kpts1 = (rng.random((n1, 2)) * np.array([640, 480])).astype(np.float32)
    matches = -np.ones(n0, np.int64)
    pick = rng.choice(n0, 20, replace=False)
    matches[pick] = rng.choice(n1, 20, replace=False)
    conf = rng.random(n0).astype(np.float32)
    n_scores0, idx0 = h["sort_scores"](scores0)
    n_kpts0 = h["sort_feature_Or_descriptors"](kpts0, idx0)
    n_scores1, idx1 = h["sort_scores"](scores1)
    n_kpts1 = h["sort_feature_Or_descriptors"](kpts1, idx1)
    n_matches = h["sort_feature_Or_descriptors"](matches, idx0)
    n_conf = h["sort_feature_Or_descriptors"](conf, idx0)
    n_matches = h["get_correct_new_matching_index"](n_matches, idx1)
    np.savez(os.path.join(OUT, "resort_ref.npz"), scores0=scores0, scores1=scores1, kpts0=kpts0, kpts1=kpts1,
             matches=matches, conf=conf, n_scores0=n_scores0, idx0=idx0, n_kpts0=n_kpts0, n_scores1=n_scores1,
             idx1=idx1, n_kpts1=n_kpts1, n_matches=n_matches, n_conf=n_conf)


if __name__ == "__main__":
    which = set(sys.argv[1:])
    if not which or "nearest_embed" in which:
        gen_nearest_embed()
        gen_nearest_embed_near_ties()
    if not which or "mutual_block" in which:
        gen_mutual_block()
    if not which or "text" in which:
        gen_text_formats()
    if not which or "resort" in which:
        gen_resort()
    print("fixtures written to", OUT)
