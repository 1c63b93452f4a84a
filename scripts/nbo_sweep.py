"""Factorisation time of config 4 against the panel-width policy (JAICOV_NBO*), one process per setting."""
import sys, os, subprocess
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(here))
    from bundle_adjustment_amd import engine, scene
    import pickle
    cache = "/tmp/cfg4.pkl"
    if os.path.exists(cache):
        fp = pickle.load(open(cache, "rb"))
    else:
        fp = scene.config("cfg4"); pickle.dump(fp, open(cache, "wb"))
    eng = engine.Engine(fp); eng.set_parameters(fp.values); s2 = fp.sigma2apriori
    best = None
    for it in range(4):
        eng.build(s2, 0.0); dx = eng.solve(False); eng.update(dx)
        tm = eng.timings()
        if it > 0: best = tm if best is None or tm["factor"] < best["factor"] else best
    print(f"{os.environ.get('JAICOV_NBO','512'):>5} big>{os.environ.get('JAICOV_NBO_BIG_ROWS','-'):>6} small<={os.environ.get('JAICOV_NBO_SMALL_ROWS','-'):>6} tail={os.environ.get('JAICOV_TAIL_ROWS','-'):>6}: factor={best['factor']:.2f} total={best['total']:.2f}", flush=True)
    eng.close()
else:
    for nbo, big, small, tail in [(512, None, None, None), (512, 8192, None, None), (512, 10240, None, None), (512, 6144, None, None),
                                 (512, 8192, 3072, None), (512, None, 3072, None), (512, 8192, None, 8192), (256, 6144, None, None)]:
        env = dict(os.environ, JAICOV_NBO=str(nbo))
        if big: env["JAICOV_NBO_BIG_ROWS"] = str(big)
        if small: env["JAICOV_NBO_SMALL_ROWS"] = str(small)
        if tail: env["JAICOV_TAIL_ROWS"] = str(tail)
        subprocess.run([sys.executable, __file__, "child"], env=env, check=False)
