import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import visual_slam_amd as vsm
from visual_slam_amd import synth
m = vsm.Matcher(0)
m.set_timing(True)
for n in (2048, 8192, 16384):
    a, b = synth.sift_pair(n, n, seed=3)
    s0, s1 = m.upload_u8(a), m.upload_u8(b)
    for _ in range(5): m.match_pair(s0, s1, ratio=0.8, mutual=True)
    t = time.perf_counter()
    for _ in range(50): g = m.match_pair(s0, s1, ratio=0.8, mutual=True)
    dt = (time.perf_counter() - t) / 50
    print(n, "call %.3f ms" % (dt * 1e3), "query kernel %.3f ms" % m.last_kernel_ms(), "verify %.3f" % m.last_verify_ms(), len(g[0]))
