import cProfile, pstats, sys, os, io
sys.argv = ["api_roundtrip.py", "200"]
sys.path.insert(0, os.getcwd())
pr = cProfile.Profile()
pr.enable()
exec(open("tools/api_roundtrip.py").read())
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
