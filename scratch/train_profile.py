"""host-side profile of train_one_epoch (where do 26 ms/step go when the GPU needs 10?)"""
import cProfile, pstats, sys, io
sys.argv = ["stability2.py", "2"]
src = open("scratch/stability2.py").read().replace("for ep in range(12):", "for ep in range(2):")
pr = cProfile.Profile()
code = compile(src.replace("out = train.train_one_epoch(", "pr.enable() if ep == 1 else None\n    out = train.train_one_epoch("), "stab", "exec")
exec(code, {"pr": pr, "__name__": "__main__"})
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
