"""List-scheduling model (scripts/flow_model_split.py) with a pool of kc CUs that run ONE tile workgroup each (step 17 us instead of 31.5) and take the
near-diagonal tasks (rows <= column + span): VERDICT r4 next 1 (ii), "critical-lane CUs", evaluated on the model before building anything.
First block: the pool takes the whole task of a near-diagonal tile -- it drowns (a tile of a late column has 100 steps).  Second block: the pool
takes only the last L steps of those tiles, the rest of their range is one more partial-sum task of the ordinary pool.
python scripts/flow_model_critical.py > profiles/r05_flow_model_critical.txt   (CPU, two minutes)"""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import flow_model as fm, numpy as np, heapq
import flow_model_split as fs
print(fm.P)
nb=118; rows=119
def simulate(nb, rows, tasks, p, kc, span, stepc, finc):
    done = np.full((rows + 1, nb), np.inf); part = {}; psum={}
    diag_done = np.full(nb, np.inf); inv_done = np.full(nb, np.inf)
    free = [0.0] * (p["WGS"]-2*kc); heapq.heapify(free)
    freec = [0.0]*max(kc,1); heapq.heapify(freec)
    diag_done[0] = 30.0; inv_done[0] = diag_done[0] + p["INV"]
    state = {"at": 0}
    def chain_to(c):
        while state["at"] < c:
            n = state["at"] + 1
            t = max(diag_done[n - 1], part[(n, n - 1)], part[(n, n)])
            done[n][n - 1] = t + 17.0
            diag_done[n] = t + p["CHAIN"]; inv_done[n] = diag_done[n] + p["INV"]
            state["at"] = n
    end=0.0
    for tk in tasks:
        i,j=tk[1],tk[2]
        crit = kc>0 and tk[0]=='M' and i<=j+span
        h = freec if crit else free
        step = stepc if crit else p["STEP"]; fin_c = finc if crit else p["FINISH"]
        t = heapq.heappop(h)
        if tk[0]=='P':
            _,i,j,k0,k1=tk
            for k in range(k0,k1):
                if done[i][k] == np.inf or done[j][k] == np.inf: chain_to(min(k + 1, nb - 1))
                t = max(t, done[i][k], done[j][k]) + step
            t += p["CSTORE"]; psum.setdefault((i,j),[]).append(t)
        else:
            _,i,j,k0,k1,fin,npart=tk
            t += p["CLOAD"]
            for k in range(k0, k1):
                if done[i][k] == np.inf or done[j][k] == np.inf: chain_to(min(k + 1, nb - 1))
                t = max(t, done[i][k], done[j][k]) + step
            for q in psum.get((i,j),[]): t=max(t,q)+p["CLOAD"]
            if fin:
                chain_to(j)
                t = max(t + p["CSTORE"], inv_done[j]) + fin_c; done[i][j] = t
            else:
                t += p["CSTORE"]
                if (i == j and k1 == max(j - 1, 0)) or (i == j + 1 and k1 == j): part[(i, j)] = t
        end=max(end,t); heapq.heappush(h,t)
    chain_to(nb-1)
    return max(end, diag_done[nb-1])
t=fs.schedule(nb,rows,2,59,'own')
P=dict(fm.P)
for kc,span in ((0,0),(8,2),(8,4),(16,2),(16,4),(16,8),(32,8)):
    for stepc,finc in ((17.0, P["FINISH"]),(17.0,P["FINISH"]*0.6)):
        print('kc',kc,'span',span,'stepc',stepc,'finc %.0f'%finc,'span %.2f ms'%(simulate(nb,rows,t,P,kc,span,stepc,finc)/1000))
print("--- critical pool takes only the last L steps of near-diagonal tiles")
def schedule2(nb, rows, jf, span, L, jc0=4):
    cols=[[] for _ in range(nb)]; extra=[[] for _ in range(nb)]
    for j in range(nb):
        for i in range(j, rows):
            k1, fin = fm.base_task(i,j,nb)
            a=0; npart=0; cuts=[]
            if j>=jf and k1>=16: cuts.append(k1//2)
            if span>0 and i<=j+span and j>=jc0 and k1-L>(cuts[-1] if cuts else 0)+2: cuts.append(k1-L)
            for c in cuts:
                extra[j].append(('P',i,j,a,c)); a=c; npart+=1
            cols[j].append(('M',i,j,a,k1,fin,npart))
    tasks=[]
    for j in range(nb): tasks+=extra[j]+cols[j]
    return tasks
for kc,span,L in ((0,0,0),(4,2,4),(8,2,4),(8,2,8),(8,3,6),(8,4,4),(16,4,8),(16,8,4)):
    t=schedule2(nb,rows,59,span,L)
    for stepc,finc in ((17.0, P["FINISH"]),(17.0,P["FINISH"]*0.6),(31.5,P["FINISH"])):
        print('kc',kc,'span',span,'L',L,'stepc',stepc,'finc %.0f'%finc,'tasks',len(t),'span %.2f ms'%(simulate(nb,rows,t,P,kc,span,stepc,finc)/1000))
