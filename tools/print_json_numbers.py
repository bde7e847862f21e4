import json,sys
m=json.load(open(sys.argv[1]))
def walk(x,pre=""):
    for k,v in x.items():
        if isinstance(v,(int,float)): print(pre+k, v)
        elif isinstance(v,dict): walk(v,pre+k+".")
walk(m)
