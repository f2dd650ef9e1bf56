import sys, json
for ln in sys.stdin:
    if not ln.startswith("{"): continue
    r = json.loads(ln)
    bound = 1e-4 * max(1, r["n"] / 1024)
    print(r["model"], r["n"], "seed", r["seed"], "cond %.1e nll %.1f  nll rel %.2e (bound %.1e %s)  factor err: nll %+.4f quad %+.4f logdet %+.4f  gram err %+.4f" % (r["cond"], r["nll64"], r["err_total_rel_nll"], bound, "ok" if r["err_total_rel_nll"] <= bound else "FAIL", r["err_factor_abs"], r["err_factor_quad"], r["err_factor_logdet"], r["err_gram_abs"]))
