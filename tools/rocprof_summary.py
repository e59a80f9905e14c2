#!/usr/bin/env python3
"""Turn a rocprofv3 result database (rocpd sqlite, the default output of
`rocprofv3 --kernel-trace --stats`) into a small text summary for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_x/em_results.db > profiles/rNN_x.txt
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    print(f'# rocprofv3 --kernel-trace --stats summary of {path}')
    print('# kernel | calls | total_us | avg_us | pct')
    for name, calls, total, avg, pct in cur.execute(
            'select name,total_calls,total_duration,average,percentage from top_kernels '
            'order by total_duration desc limit 12'):
        print(f'{name} | {calls} | {total:.1f} | {avg:.2f} | {pct:.2f}')
    print('# per-dispatch resources of the pbbss kernels: name | n | avg_ns | min_ns | max_ns | '
          'grid | wg | lds_bytes | scratch_bytes_per_lane | vgpr | agpr | sgpr')
    q = ('select name,count(*),avg(duration),min(duration),max(duration),grid_x,workgroup_x,'
         'lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count from kernels '
         "where name like '%pbbss%' group by name,grid_x,lds_size order by avg(duration) desc")
    for r in cur.execute(q):
        print(' | '.join(str(round(x) if isinstance(x, float) else x) for x in r))


if __name__ == '__main__':
    main(sys.argv[1])
