"""Summarise rocprofv3's default rocpd SQLite output: per kernel mean duration, and mean of each PMC counter."""
import collections, glob, sqlite3, sys
paths = sys.argv[1:] or sorted(glob.glob("gpurun_out/**/*_results.db", recursive=True))[-1:]
for path in paths:
    c = sqlite3.connect(path)
    names = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    # tables carry a per-run suffix (rocpd_kernel_dispatch_<guid>): the shortest name that starts with the wanted one
    t = lambda n: min((x for x in names if x == n or x.startswith(n + "_")), key=len)
    if not any(x.startswith("rocpd_info_kernel_symbol") for x in names):
        print("==", path, "(no kernel tables: skipped)")
        continue
    sym = dict(c.execute(f"select id, kernel_name from {t('rocpd_info_kernel_symbol')}"))
    disp = {}
    dur = collections.defaultdict(list)
    cols = [r[1] for r in c.execute(f"pragma table_info({t('rocpd_kernel_dispatch')})")]
    for r in c.execute(f"select * from {t('rocpd_kernel_dispatch')}"):
        d = dict(zip(cols, r))
        name = sym.get(d.get("kernel_id"), "?")
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void suave::", "").replace("suave::", "")
        grid = d.get("grid_size_x", 0) // max(d.get("workgroup_size_x", 1), 1)
        disp[d["id"]] = (short, grid, d.get("event_id"))
        dur[(short, grid)].append(d["end"] - d["start"])
    print("==", path)
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if "suave" in k[0] or "k_" in k[0] or "gemm" in k[0]:
            print(f"{k[0][:60]:60s} blocks={k[1]:7d} n={len(v):4d} mean_us={sum(v)/len(v)/1e3:10.2f}")
    try:
        pmc = dict(c.execute(f"select id, symbol from {t('rocpd_info_pmc')}"))
        ev2disp = {v[2]: k for k, v in disp.items()}
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        pcols = [r[1] for r in c.execute(f"pragma table_info({t('rocpd_pmc_event')})")]
        for r in c.execute(f"select * from {t('rocpd_pmc_event')}"):
            d = dict(zip(pcols, r))
            did = ev2disp.get(d.get("event_id"))
            if did is None:
                continue
            acc[disp[did][:2]][pmc.get(d.get("pmc_id"), "?")].append(d.get("value", 0.0))
        for k, dd in acc.items():
            if any(w in k[0] for w in ("decode", "gemm", "loss", "sample", "k_")):
                print(f"{k[0][:60]} blocks={k[1]}")
                for cn, v in sorted(dd.items()):
                    print(f"    {cn:32s} {sum(v)/len(v):18.1f} (n={len(v)})")
    except Exception as e:  # noqa: BLE001
        print("no pmc:", e)
