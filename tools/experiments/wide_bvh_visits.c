// scratch experiment: visits per ray for 4-wide sorted vs 8-wide sorted vs 8-wide octant-order traversal (quantised boxes)
#include "/root/repo/oracle/crh_oracle.c"
#include <math.h>

#define MAXW 8
typedef struct { float org[3]; uint32_t e[3]; int n; uint8_t qlo[MAXW][3], qhi[MAXW][3]; int32_t child[MAXW]; /* >=0 wide node, <0: ~leaf prim pos */ } wnode;
static wnode* WN; static uint32_t nWN, capWN;
static const bnode* BN; static const uint32_t* IDX;
static int WIDTH = 8, SLOTMODE = 0;

static uint32_t wn_new(void){ if(nWN==capWN){capWN*=2; WN=realloc(WN,sizeof(wnode)*capWN);} return nWN++; }

static uint32_t collapse_w(uint32_t bi)
{
  uint32_t me = wn_new();
  uint32_t kids[MAXW]; int nk=0;
  const bnode* b=&BN[bi];
  if (b->left<0) { kids[nk++]=bi; }
  else {
    kids[nk++]=b->left; kids[nk++]=b->right;
    for(;;){
      if(nk>=WIDTH) break;
      int best=-1; float ba=-1;
      for(int k=0;k<nk;++k){ const bnode* c=&BN[kids[k]]; if(c->left>=0){ float a=aabb_harea(&c->box); if(a>ba){ba=a;best=k;} } }
      if(best<0) break;
      const bnode* c=&BN[kids[best]];
      uint32_t l=c->left,r=c->right;
      // keep order: replace best by l, insert r after
      for(int k=nk;k>best+1;--k) kids[k]=kids[k-1];
      kids[best]=l; kids[best+1]=r; nk++;
    }
  }
  // slot assignment
  uint32_t slot[MAXW]; int used[MAXW]={0}; int ns=nk;
  if (SLOTMODE==1 && WIDTH==8) {
    float cen[3]={0,0,0}; aabb u; aabb_empty(&u);
    for(int k=0;k<nk;++k) aabb_grow(&u,&BN[kids[k]].box);
    for(int a=0;a<3;++a) cen[a]=0.5f*(u.mn[a]+u.mx[a]);
    int assigned[MAXW]; for(int k=0;k<nk;++k) assigned[k]=0;
    for(int s=0;s<8;++s) slot[s]=0xFFFFFFFFu;
    for(int it=0;it<nk;++it){
      float bc=-3e38f; int bk=-1,bs=-1;
      for(int k=0;k<nk;++k) if(!assigned[k]) for(int s=0;s<8;++s) if(!used[s]){
        const aabb* cb=&BN[kids[k]].box; float c=0;
        for(int a=0;a<3;++a){ float d=0.5f*(cb->mn[a]+cb->mx[a])-cen[a]; c+= ((s>>a)&1)? d : -d; }
        if(c>bc){bc=c;bk=k;bs=s;}
      }
      assigned[bk]=1; used[bs]=1; slot[bs]=kids[bk];
    }
    ns=8;
  } else { for(int k=0;k<nk;++k) slot[k]=kids[k]; }
  wnode w; memset(&w,0,sizeof w); w.n=ns;
  float lo[3]={3e38f,3e38f,3e38f},hi[3]={-3e38f,-3e38f,-3e38f};
  for(int k=0;k<ns;++k) if(slot[k]!=0xFFFFFFFFu){ const aabb* cb=&BN[slot[k]].box; for(int a=0;a<3;++a){ if(cb->mn[a]<lo[a])lo[a]=cb->mn[a]; if(cb->mx[a]>hi[a])hi[a]=cb->mx[a]; } }
  for(int a=0;a<3;++a){ w.org[a]=lo[a]; w.e[a]=crh_quant_exp(hi[a]-lo[a]); }
  for(int k=0;k<ns;++k){
    if(slot[k]==0xFFFFFFFFu){ for(int a=0;a<3;++a){w.qlo[k][a]=255; w.qhi[k][a]=0;} w.child[k]=0x7fffffff; continue; }
    const aabb* cb=&BN[slot[k]].box;
    for(int a=0;a<3;++a){ w.qlo[k][a]=crh_quant_lo(cb->mn[a],w.org[a],w.e[a]); w.qhi[k][a]=crh_quant_hi(cb->mx[a],w.org[a],w.e[a]); }
  }
  WN[me]=w;
  for(int k=0;k<ns;++k){
    if(slot[k]==0xFFFFFFFFu) continue;
    const bnode* c=&BN[slot[k]];
    int32_t ch;
    if(c->left<0) ch = ~(int32_t)IDX[c->lo]; else ch=(int32_t)collapse_w(slot[k]);
    WN[me].child[k]=ch;
  }
  return me;
}

static uint64_t g_nodes,g_tris;
static const orc_ctx* CTX;
static float TRIV[1<<21][9];

static int trav(v3 o, v3 d, float tmax, int mode /*0 sorted,1 octant*/)
{
  int32_t stack[256]; int sp=0;
  float ix=inv_dir(d.x),iy=inv_dir(d.y),iz=inv_dir(d.z);
  float best=tmax; int found=-1;
  int r = (d.x>=0?1:0)|(d.y>=0?2:0)|(d.z>=0?4:0);
  int32_t cur=0;
  for(;;){
    if(cur<0){
      uint32_t p=~cur; g_tris++;
      qtri q; for(int k=0;k<3;++k){ const float* pp=&CTX->pos[3*CTX->tri[4*p+k]]; q.f[4*k]=pp[0];q.f[4*k+1]=pp[1];q.f[4*k+2]=pp[2]; }
      float t,u,v; if(tri_test(&q,o,d,best,&t,&u,&v)){best=t;found=p;}
    } else {
      const wnode* w=&WN[cur]; g_nodes++;
      float inv[3]={ix,iy,iz}; float oo[3]={o.x,o.y,o.z};
      float key[MAXW]; int ord[MAXW]; int nh=0;
      for(int k=0;k<w->n;++k){
        float tmin=0,tmx=best;
        for(int a=0;a<3;++a){
          float st=crh_quant_step(w->e[a]);
          float p0=w->org[a]+w->qlo[k][a]*st, p1=w->org[a]+w->qhi[k][a]*st;
          float t0=(p0-oo[a])*inv[a], t1=(p1-oo[a])*inv[a];
          if(t0>t1){float x=t0;t0=t1;t1=x;}
          if(t0>tmin)tmin=t0; if(t1<tmx)tmx=t1;
        }
        if(w->qlo[k][0]>w->qhi[k][0]) continue;
        if(tmin<=tmx){ key[nh]= mode? (float)(k ^ r) : tmin; ord[nh]=k; nh++; }   /* octant: visit decreasing (k^r) -> sort ascending on -(k^r) */
      }
      if(mode) for(int i=0;i<nh;++i) key[i]=-key[i];
      for(int i=1;i<nh;++i){ float kk=key[i]; int oo2=ord[i]; int j=i; while(j>0&&key[j-1]>kk){key[j]=key[j-1];ord[j]=ord[j-1];--j;} key[j]=kk;ord[j]=oo2; }
      if(nh>0){ for(int j=nh-1;j>=1;--j) stack[sp++]=w->child[ord[j]]; cur=w->child[ord[0]]; continue; }
    }
    if(sp==0)break; cur=stack[--sp];
  }
  return found;
}

int main(int argc,char**argv)
{
  uint32_t n = argc>1? atoi(argv[1]) : 1000000;
  // gen_scene equivalent (not identical RNG): random triangles
  orc_ctx* c=orc_create();
  float* pos=malloc(sizeof(float)*9*n); int32_t* tri=malloc(sizeof(int32_t)*4*n); float* nrm=calloc(9*n,sizeof(float));
  srand48(1); float r=1.5f*powf((float)n,-1.f/3.f);
  for(uint32_t t=0;t<n;++t){ float cx=drand48()*2-1,cy=drand48()*2-1,cz=drand48()*2-1;
    pos[9*t]=cx;pos[9*t+1]=cy;pos[9*t+2]=cz;
    for(int k=1;k<3;++k){ pos[9*t+3*k]=cx+(drand48()*2-1)*r; pos[9*t+3*k+1]=cy+(drand48()*2-1)*r; pos[9*t+3*k+2]=cz+(drand48()*2-1)*r; }
    tri[4*t]=3*t;tri[4*t+1]=3*t+1;tri[4*t+2]=3*t+2;tri[4*t+3]=0; }
  c->pos=pos;c->tri=tri;c->nV=3*n;c->nT=n; CTX=c;
  aabb* pb=malloc(sizeof(aabb)*n); for(uint32_t t=0;t<n;++t) tri_box(c,t,&pb[t]);
  float* cen=malloc(sizeof(float)*3*n); for(uint32_t t=0;t<n;++t) for(int a=0;a<3;++a) cen[3*t+a]=(pb[t].mn[a]+pb[t].mx[a])*0.5f;
  builder B; B.pb=pb;B.cen=cen;B.leaf_max=1; B.idx=malloc(4*n);B.tmp=malloc(4*n); for(uint32_t t=0;t<n;++t)B.idx[t]=t; B.cap=1024;B.nbn=0;B.bn=malloc(sizeof(bnode)*B.cap);
  build_rec(&B,0,n,0); BN=B.bn; IDX=B.idx;
  fprintf(stderr,"binary nodes %u\n",B.nbn);
  int nr=200000;
  float* rays=malloc(sizeof(float)*6*nr);
  for(int i=0;i<nr;++i){ // secondary-like: origin on a random triangle, random direction
    uint32_t t=lrand48()%n; float dx,dy,dz,l; do{dx=drand48()*2-1;dy=drand48()*2-1;dz=drand48()*2-1;l=dx*dx+dy*dy+dz*dz;}while(l>1||l<1e-4); l=sqrtf(l);
    rays[6*i]=pos[9*t]+1e-4f*dx;rays[6*i+1]=pos[9*t+1]+1e-4f*dy;rays[6*i+2]=pos[9*t+2]+1e-4f*dz;rays[6*i+3]=dx/l;rays[6*i+4]=dy/l;rays[6*i+5]=dz/l; }
  struct { int width, slotmode, travmode; const char* name; } cfg[] = { {4,0,0,"4-wide sorted (greedy collapse)"}, {8,0,0,"8-wide sorted"}, {8,1,1,"8-wide octant slots + octant order"}, {8,1,0,"8-wide octant slots, sorted order"} };
  for(int ci=0;ci<4;++ci){
    WIDTH=cfg[ci].width; SLOTMODE=cfg[ci].slotmode; capWN=1024;nWN=0;WN=malloc(sizeof(wnode)*capWN);
    collapse_w(0);
    g_nodes=g_tris=0; long hits=0; double ts=0;
    for(int i=0;i<nr;++i){ int f=trav(crh_mk3(rays[6*i],rays[6*i+1],rays[6*i+2]),crh_mk3(rays[6*i+3],rays[6*i+4],rays[6*i+5]),3e38f,cfg[ci].travmode); hits+=f>=0; }
    printf("%-40s nodes %8u  visits/ray %.2f  tris/ray %.2f  hit %.3f\n",cfg[ci].name,nWN,(double)g_nodes/nr,(double)g_tris/nr,(double)hits/nr);
    free(WN);
  }
  return 0;
}
