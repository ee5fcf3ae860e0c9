import csv,glob,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for path in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(path)):
        kn=r['Kernel_Name']
        if 'conv_' not in kn: continue
        import re
        m=re.search(r'(conv_\w+)<([^>]*)>',kn)
        key=(m.group(1)+'<'+m.group(2)+'>') if m else kn[:60]
        key=key+' grid='+r.get('Grid_Size','?')
        agg[key][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(key,r['Counter_Name'])]+=1
for k,v in agg.items():
    print(k)
    for c,val in sorted(v.items()): print('   %-28s %.4g'%(c,val/cnt[(k,c)]))
