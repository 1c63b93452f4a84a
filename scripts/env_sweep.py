"""Times config 4 under a list of environment settings, one process each (exploration helper).
usage: env_sweep.py "A=1 B=2" "A=3" ..."""
import sys, os, subprocess, pickle
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(here))
    from bundle_adjustment_amd import engine, scene
    cache = "/tmp/cfg4.pkl"
    if os.path.exists(cache):
        fp = pickle.load(open(cache, "rb"))
    else:
        fp = scene.config("cfg4"); pickle.dump(fp, open(cache, "wb"))
    eng = engine.Engine(fp); eng.set_parameters(fp.values); s2 = fp.sigma2apriori
    best = None
    for it in range(5):
        eng.build(s2, 0.0); dx = eng.solve(False); eng.update(dx)
        tm = eng.timings()
        if it > 0: best = tm if best is None or tm["total"] < best["total"] else best
    print(f"{sys.argv[2]:40s}: assembly={best['assembly']:.2f} factor={best['factor']:.2f} solve={best['solve']:.2f} total={best['total']:.2f}", flush=True)
    eng.close()
else:
    for setting in sys.argv[1:]:
        env = dict(os.environ)
        for kv in setting.split():
            if "=" in kv:
                k, v = kv.split("=", 1); env[k] = v
        subprocess.run([sys.executable, __file__, "--child", setting or "(default)"], env=env, check=False)
