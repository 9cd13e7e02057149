import json,sys
for f in sys.argv[1:]:
    r=json.load(open(f))
    tot=sum(x['total_ms'] for x in r)/3
    print(f, 'gpu ms/step', round(tot,3), 'launches', sum(x['calls'] for x in r)//3)
    for x in r:
        n=x['name']
        if ('K327680' in n) or 'k_post_prep' in n or 'splitk' in n and x['total_ms']/3>0.005:
            print('   %8.1f us x%d  %s'%(x['total_ms']/x['calls']*1000, x['calls']//3, n))
