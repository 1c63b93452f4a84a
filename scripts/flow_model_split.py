"""List-scheduling model of the dataflow factorisation WITH split update ranges (round 5): the model of scripts/flow_model.py (round 4 costs) plus
partial-sum tasks -- independent of the dependency front, placed at the head of their column ("own") or a few columns behind the last column
they need -- whose sums the tile s own task adds at the end.  It is what said, before anything was built, that cutting the late columns ranges
shortens the span where no reordering of whole tasks did (python scripts/flow_model_split.py > profiles/r05_flow_model_split.txt; CPU, a minute).
Measured afterwards at 118 block columns: two pieces from column 59 on -1.1 ms (model: -0.7); four pieces WORSE than none (model: better) --
the model has no cost for a waiting task that holds a slot, which is what four pieces run into."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import flow_model as fm, numpy as np, heapq
nb=118; rows=119
def schedule(nb, rows, m=1, jf=999, place='own'):
    """split-K: tile (i, j >= jf): K range cut into m pieces; the first m-1 are independent partial-sum tasks (kind 'P'), placed either at the head of
    column j's tasks ('own') or right behind the last column they need ('early'); the main task does the last piece, adds the partial sums, finishes"""
    cols=[[] for _ in range(nb)]; extra=[[] for _ in range(nb)]
    for j in range(nb):
        for i in range(j, rows):
            k1, fin = fm.base_task(i,j,nb)
            a=0; npart=0
            if j>=jf and k1>=8*m:
                cuts=[k1*q//m for q in range(1,m)]
                for c in cuts:
                    tsk=('P',i,j,a,c)
                    if place=='own': extra[j].append(tsk)
                    else: extra[min(c-1+place, j)].append(tsk)
                    a=c; npart+=1
            cols[j].append(('M',i,j,a,k1,fin,npart))
    tasks=[]
    for j in range(nb):
        if place=='own': tasks+=extra[j]+cols[j]
        else: tasks+=cols[j]+extra[j]
    return tasks
def simulate(nb, rows, tasks, p):
    done = np.full((rows + 1, nb), np.inf); part = {}; psum={}
    diag_done = np.full(nb, np.inf); inv_done = np.full(nb, np.inf)
    free = [0.0] * p["WGS"]; heapq.heapify(free)
    diag_done[0] = 30.0; inv_done[0] = diag_done[0] + p["INV"]
    state = {"at": 0}
    def chain_to(c):
        while state["at"] < c:
            n = state["at"] + 1
            t = max(diag_done[n - 1], part[(n, n - 1)], part[(n, n)])
            done[n][n - 1] = t + 17.0
            diag_done[n] = t + p["CHAIN"]; inv_done[n] = diag_done[n] + p["INV"]
            state["at"] = n
    end=0.0; busy=0.0
    for tk in tasks:
        t0 = t = heapq.heappop(free)
        if tk[0]=='P':
            _,i,j,k0,k1=tk
            for k in range(k0,k1):
                if done[i][k] == np.inf or done[j][k] == np.inf: chain_to(min(k + 1, nb - 1))
                t = max(t, done[i][k], done[j][k]) + p["STEP"]
            t += p["CSTORE"]; psum.setdefault((i,j),[]).append(t)
        else:
            _,i,j,k0,k1,fin,npart=tk
            t += p["CLOAD"]
            for k in range(k0, k1):
                if done[i][k] == np.inf or done[j][k] == np.inf: chain_to(min(k + 1, nb - 1))
                t = max(t, done[i][k], done[j][k]) + p["STEP"]
            for q in psum.get((i,j),[]): t=max(t,q)+p["CLOAD"]
            if fin:
                chain_to(j)
                t = max(t + p["CSTORE"], inv_done[j]) + p["FINISH"]; done[i][j] = t
            else:
                t += p["CSTORE"]
                if (i == j and k1 == max(j - 1, 0)) or (i == j + 1 and k1 == j): part[(i, j)] = t
        end=max(end,t); heapq.heappush(free,t)
    chain_to(nb-1)
    return max(end, diag_done[nb-1])
P=dict(fm.P)
for m,jf,place in ((1,999,'own'),(2,88,'own'),(2,64,'own'),(3,88,'own'),(4,88,'own'),(4,64,'own'),(2,88,3),(4,88,3),(4,64,3),(4,64,10),(8,64,10)):
    t=schedule(nb,rows,m,jf,place)
    for chain,inv in ((60,25),(40,5)):
        q=dict(P); q['CHAIN']=chain; q['INV']=inv
        print('m',m,'from',jf,place,'chain',chain,'tasks',len(t),'span %.2f'%(simulate(nb,rows,t,q)/1000))
