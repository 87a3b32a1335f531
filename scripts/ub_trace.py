import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ub = [r for r in rows if 'ub_layer' in r['Kernel_Name'] or 'ub_cond' in r['Kernel_Name']]
# sequences of 22 launches (1 cond + 21 layers)
seq = []
cur = []
for r in ub:
    if 'ub_cond' in r['Kernel_Name']:
        if cur: seq.append(cur)
        cur = []
    cur.append(r)
if cur: seq.append(cur)
seq = [s for s in seq if len(s) == len(seq[-1])]
n = len(seq[-1])
print('forwards', len(seq), 'launches per forward', n)
tot = 0
for i in range(n):
    d = [int(s[i]['End_Timestamp']) - int(s[i]['Start_Timestamp']) for s in seq]
    g = [int(s[i]['Start_Timestamp']) - int(s[i-1]['End_Timestamp']) for s in seq] if i else [0]
    m = sum(d)/len(d)/1000; tot += m
    print(i, seq[-1][i]['Kernel_Name'][:30], 'grid', seq[-1][i].get('Grid_Size_X', seq[-1][i].get('Grid_Size')), 'lds', seq[-1][i].get('LDS_Block_Size'), '%.1f us' % m, 'gap %.1f us' % (sum(g)/len(g)/1000))
print('sum %.1f us' % tot, 'span %.1f us' % (sum(int(s[-1]['End_Timestamp']) - int(s[0]['Start_Timestamp']) for s in seq)/len(seq)/1000))
