"""What bench.py prints and what it files away.

The driver keeps only a few KB of the ONE JSON line (round 5's 15 KB line survived as key names and two truncated tails), so the
line itself is the contract's keys + `roofline` + `cpu_baseline` + `parity` + one SMALL object per secondary figure
({value, unit, frac, ms, abs_dpsnr_db | bitwise ...}); every object in full -- workloads, samples, notes, per-stage tables --
goes to `bench_full.json` next to bench.py, and the line names that file."""
import json
import os

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config")
LINE_BUDGET_BYTES = 4096


def _sig(x, digits=5):
    """Floats to `digits` significant digits (the full file keeps every bit); everything else untouched."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return x
    return float(f"{x:.{digits}g}")


def _pick(obj, *keys):
    return {k: _sig(obj[k]) for k in keys if isinstance(obj, dict) and k in obj and obj[k] is not None}


def _frac(obj):
    roof = obj.get("roofline") if isinstance(obj, dict) else None
    if not isinstance(roof, dict) or "frac" not in roof:
        return {}
    return {"frac": _sig(roof["frac"]), **({"bound": roof.get("bound")} if roof.get("bound") != "mfma" else {})}   # mfma unless it says so


def _small(obj, *keys):
    """A secondary object's compact form: an error stays an error; otherwise the requested scalars + its roofline fraction +
    the parity figure that applies (|dPSNR| against the oracle, or a bitwise verdict)."""
    if not isinstance(obj, dict):
        return obj
    if "error" in obj:
        return {"error": str(obj["error"])[:200], **_pick(obj, "failed_ranks")}
    out = _pick(obj, "value", "unit", *keys)
    out.update(_frac(obj))
    par = obj.get("parity")
    if isinstance(par, dict):
        out.update(_pick(par, "abs_dpsnr_db", "rays_over_1e-4"))
    cpu = obj.get("cpu_baseline")
    if isinstance(cpu, dict) and "value" in cpu:
        out["cpu"] = _cpu(cpu, out.get("unit"))
    return out


def _cpu(cpu, unit=None):
    """A secondary CPU leg on the line: rate and threads (its unit only where it is not the object's own; kind / sample: full file)."""
    c = _pick(cpu, "value", "cores")
    if cpu.get("unit") != unit:
        c["unit"] = cpu.get("unit")
    return c


def _mesh(m):
    if not isinstance(m, dict) or "error" in m:
        return _small(m)
    out = _pick(m, "end_to_end_s")
    gq = m.get("grid_query")
    if isinstance(gq, dict):
        g = _pick(gq, "points", "ms")
        g.update(_frac(gq))
        if isinstance(gq.get("parity"), dict):
            g.update(_pick(gq["parity"], "max_abs_dsigma_over_scale"))
        if isinstance(gq.get("cpu_baseline"), dict):
            g["cpu"] = _cpu(gq["cpu_baseline"])
        if isinstance(gq.get("at_reference_batch_1024"), dict):
            g["at_reference_batch_1024"] = _pick(gq["at_reference_batch_1024"], "value", "unit", "calls", "ms_per_call", "full_grid_s",
                                                 "same_sigma_as_one_call")
        for k in ("all_gather", "scaling"):
            if isinstance(gq.get(k), dict):
                g[k] = {kk: _sig(v) for kk, v in gq[k].items() if not isinstance(v, (dict, list, str))}
        out["grid_query"] = g
    mc = m.get("marching_cubes")
    if isinstance(mc, dict):
        c = _pick(mc, "vertices", "faces", "ms_avg")
        c.update({short: mc[k] for k, short in (("iso_equals_numpy_fp32", "iso_np_exact"), ("bitwise_identical_to_oracle", "bitwise")) if k in mc})
        c.update(_frac(mc))
        if isinstance(mc.get("roofline"), dict):
            c.update(_pick(mc["roofline"], "traffic"))
        if isinstance(mc.get("cpu_baseline"), dict):
            c["cpu"] = _cpu(mc["cpu_baseline"])
        out["marching_cubes"] = c
    sh = m.get("sharded")
    if isinstance(sh, dict) and isinstance(sh.get("strategies"), dict):
        st = sh["strategies"]
        out["sharded"] = {"default": sh.get("default"),
                          **{k + "_ms": _sig(v.get("ms_end_to_end")) for k, v in st.items() if isinstance(v, dict)},
                          "meshes_equal_single_grid": all(v.get("faces_and_normals_equal_single_grid_mesh") is True for v in st.values()
                                                          if isinstance(v, dict)),
                          "grid_all_gather_ms": _sig((sh.get("all_gather_of_the_grid") or {}).get("ms"))}
    app = m.get("appearance")
    if isinstance(app, dict):
        out["appearance"] = {}
        for k, v in app.items():
            if isinstance(v, dict):
                a = _pick(v, "end_to_end_s", "requery_rays_per_s", "obj_MBps")
                a.update(_frac(v))
                if isinstance(v.get("parity"), dict):
                    a.update(_pick(v["parity"], "abs_dpsnr_db"))
                out["appearance"][k] = a
    par = m.get("parity")
    if isinstance(par, dict) and isinstance(par.get("topology"), dict):
        topo = par["topology"]
        out["topology_128"] = {short: topo[k] for k, short in (("sign_flips_at_iso", "sign_flips"), ("cubes_cut_by_the_surface", "cut_cubes"),
                                                                 ("cubes_whose_corner_pattern_differs", "differ"), ("abs_dV", "dV"),
                                                                 ("abs_dF", "dF"), ("within_budget", "ok")) if k in topo}
    return out


def _train(t):
    if not isinstance(t, dict) or "error" in t:
        return _small(t)
    out = _small(t, "ms_per_iteration", "rays_per_iteration")
    if isinstance(t.get("roofline"), dict) and "frac_of_reference_work" in t["roofline"]:
        out["frac_ref_work"] = t["roofline"]["frac_of_reference_work"]      # `frac` counts the matrix work executed, this one the reference's autograd
    if isinstance(t.get("kernels"), dict):
        out["stages"] = {k: _pick(v, "ms", "frac") for k, v in t["kernels"].items() if isinstance(v, dict) and v.get("ms", 0) >= 0.1}
    if isinstance(t.get("shapes"), dict):
        out["shapes"] = {k: _pick(v, "ms_per_iteration", "frac", "ms_graph_replay", "frac_graph_replay") for k, v in t["shapes"].items()
                         if isinstance(v, dict)}
    return out


def compact_line(out, full_path=None):
    """The line the driver records: contract keys as they are, everything else reduced to the figures a reviewer reads."""
    line = {k: out[k] for k in CONTRACT if k in out}
    if isinstance(line.get("config"), dict):
        line["config"] = dict(line["config"])
    roof = out.get("roofline")
    if isinstance(roof, dict):
        line["roofline"] = _pick(roof, "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches", "avg_launch_ms")
        line["roofline"].setdefault("traffic", None)
        if roof.get("traffic_source"):
            line["roofline"]["traffic_source"] = str(roof["traffic_source"]).split(":")[0]
    cpu = out.get("cpu_baseline")
    if isinstance(cpu, dict):
        line["cpu_baseline"] = _pick(cpu, "value", "unit", "cores", "kind", "speedup", "port_over_reference_time")
        line["cpu_baseline"]["sample"] = str(cpu.get("sample", ""))[:110]
    par = out.get("parity")
    if isinstance(par, dict):
        line["parity"] = _pick(par, "psnr_ref_db", "abs_dpsnr_db", "max_abs_drgb", "rays", "rays_over_1e-4")
        if isinstance(par.get("rays_over_1e-4_explained"), dict):
            line["parity"]["unexplained"] = par["rays_over_1e-4_explained"].get("unexplained")
    for k in ("ranks_per_gpu", "note", "errors"):
        if k in out:
            line[k] = out[k] if k != "note" else str(out[k])[:200]
    if isinstance(out.get("rccl"), dict):
        line["rccl"] = {k: (_sig(v) if not isinstance(v, list) else [_sig(x) for x in v]) for k, v in out["rccl"].items()}
    if isinstance(out.get("scaling_detail"), dict):
        line["scaling_detail"] = {k: _sig(v) for k, v in out["scaling_detail"].items()}
    if "mesh" in out:
        line["mesh"] = _mesh(out["mesh"])
    if "buff" in out:
        line["buff"] = _small(out["buff"], "ms_per_view")
    if "eval" in out:
        line["eval"] = _small(out["eval"], "ms_per_view", "views", "dataset_psnr_db")
        ref = out["eval"].get("at_reference_chunksize") if isinstance(out["eval"], dict) else None
        if isinstance(ref, dict):
            line["eval"]["at_reference_chunksize"] = _pick(ref, "chunk_rays", "value", "same_loss_as_large_calls")
    if "tiny" in out:
        line["tiny"] = _small(out["tiny"], "ms_per_view")
        tr = out["tiny"].get("train") if isinstance(out["tiny"], dict) else None
        if isinstance(tr, dict):
            line["tiny"]["train"] = _pick(tr, "ms_per_iteration_eager", "ms_per_iteration_graph_replay", "frac_graph_replay", "frac_ref_work",
                                          "rays_per_s_graph_replay")
    if "train" in out:
        line["train"] = _train(out["train"])
    if "bf16x3" in out:
        line["bf16x3"] = _small(out["bf16x3"], "ms_per_view", "dtype", "algorithmic_tflops")
    if "drop_in" in out:
        line["drop_in"] = out["drop_in"] if not isinstance(out["drop_in"], dict) or "error" in out["drop_in"] else \
            {k: _sig(v) for k, v in out["drop_in"].items() if not isinstance(v, (dict, list)) or k == "repacks"}
    if full_path:
        line["full"] = full_path
    return line


def render(out, full_path=None):
    """JSON text of the compact line; should an object still push it past the budget, the largest secondary objects are cut to
    their first-level scalars until it fits (the full file has them anyway)."""
    line = compact_line(out, full_path)
    text = json.dumps(line, default=repr)
    order = sorted((k for k in line if k not in CONTRACT and k not in ("roofline", "cpu_baseline", "parity", "full")),
                   key=lambda k: -len(json.dumps(line[k], default=repr)))
    for k in order:
        if len(text) <= LINE_BUDGET_BYTES:
            break
        if isinstance(line[k], dict):
            line[k] = {kk: v for kk, v in line[k].items() if not isinstance(v, (dict, list))}
            text = json.dumps(line, default=repr)
    return text


def write_full(out, path):
    """Every object as measured.  Best effort: a read-only checkout must not cost the line."""
    try:
        tmp = path + ".tmp"
        with open(tmp, "w") as fh:
            json.dump(out, fh, indent=1, default=repr)
        os.replace(tmp, path)
        return True
    except OSError:
        return False
