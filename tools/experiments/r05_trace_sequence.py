import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
fin = [i for i, r in enumerate(rows) if 'final_outputs_state' in r['Kernel_Name']]
a, b = fin[-2], fin[-1]
prev = None
out = []
for r in rows[a + 1:b + 1]:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('dvo::', '')[:34]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append('%s %.1f(+%.1f)' % (n, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
print(len(out), 'launches in one alignment')
print('\n'.join(out))
