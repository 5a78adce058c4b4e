import csv,glob,sys
d=sys.argv[1]
for p in glob.glob(d+'/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        if 'copyBuffer' in r['Kernel_Name']: print('BLIT kernel dur %.3f ms'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
for p in glob.glob(d+'/**/*memory_copy_trace.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        print(r['Direction'], 'dur %.3f ms'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
