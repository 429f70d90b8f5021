import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])   # debug build, reference build
bad = (a["seeded_t"].view(np.uint32) != b["seeded_t"].view(np.uint32)) | (a["seeded_primitive_id"] != b["seeded_primitive_id"])
sp, info = a["seeded_bary_u"], a["seeded_bary_v"]
inq = sp >= 0
print("rays", bad.size, "in quad mode", int(inq.sum()), "bad", int(bad.sum()), "bad & in quad", int((bad & inq).sum()))
for name, m in (("all quad rays", inq), ("bad rays", bad & inq)):
    s, i = sp[m], info[m]
    leaf = i >= 1000
    n_act = np.where(leaf, i - 1000, i)
    print(name, ": sp hist", np.bincount(s.astype(int), minlength=30)[:30], "\n   at leaf:", int(leaf.sum()), "of", len(s), " n_act hist", np.bincount(n_act.astype(int), minlength=17)[:17])
