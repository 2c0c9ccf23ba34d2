"""Mean of every SQ counter collected by tools/pmc_pass_kernel.sh for the pass kernel, per sampled unit and per 16-row tile
(the counters of one dispatch cover 1/32 of the chip: 8 CUs = 32 SIMDs = 195.3 of the 6250 tiles of config 2)."""
import sqlite3, glob, sys
for db in sorted(glob.glob('gpurun_out/pmc_*/r_results.db')):
    con = sqlite3.connect(db); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    try:
        pm = [t for t in tabs if 'pmc_event' in t][0]; ip = [t for t in tabs if 'info_pmc' in t][0]
        kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
        names = dict(cur.execute("select id, kernel_name from %s" % ks).fetchall())
        q = "select p.name, k.kernel_id, avg(e.value), count(*) from %s e join %s p on e.pmc_id=p.id join %s k on e.event_id=k.event_id group by p.name,k.kernel_id" % (pm, ip, kd)
        for r in cur.execute(q).fetchall():
            if 'eval_kernel' in names.get(r[1], '?'):
                print("%-32s mean %12.0f  per tile %9.1f" % (r[0], r[2], r[2] / 195.3))
    except Exception as ex:
        print('ERR', db, ex)
