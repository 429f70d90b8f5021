"""Pipe classes of the pool's streams in creation order, then the strip / frame times in a clean process and after other renderers."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tauray_amd import renderer as R
ctx = R.Context(0)
print("null stream: pipe class", ctx.stream_pipe_class(None))
ss = [ctx.create_stream() for _ in range(10)]
print("ten streams in the order the pool hands them out:", [ctx.stream_pipe_class(s) for s in ss])
for s in ss: ctx.destroy_stream(s)
ss = [ctx.create_stream() for _ in range(4)]
print("four streams after the ten went back:", [ctx.stream_pipe_class(s) for s in ss])
for s in ss: ctx.destroy_stream(s)
