import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    x, y = a[k], b[k]
    if x.dtype.kind == "f":
        same = (x.view(np.uint32) == y.view(np.uint32))
    else:
        same = x == y
    bad = ~same
    if k.startswith("frame"):
        bad = bad.any(-1)
    print(f"{k}: {int(bad.sum())} of {bad.size} differ")
    if bad.any() and not k.startswith("frame"):
        idx = np.nonzero(bad)[0][:5]
        print("   first:", idx, x[idx], y[idx])
