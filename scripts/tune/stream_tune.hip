// Standalone tuning harness for the fused 7-stream iteration kernel and the 3-stream dir kernel.
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 stream_tune.hip -o stream_tune
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int NS = 10, BLOCK = 256;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s\n",hipGetErrorString(e),#x); exit(1);} }while(0)

struct P { double *x,*u; const double *g; double *gt; const double *p0; long long n; double a_acc,beta,a_trial; double *partials; unsigned *ticket; double *out; };

__device__ inline double wave_sum(double v){
#pragma unroll
  for(int off=32;off>0;off>>=1) v+=__shfl_down(v,off,64);
  return v; }

template<bool NT> __device__ inline d2 ld(const double* p,long long i){ const d2* q=reinterpret_cast<const d2*>(p)+i; if(NT) return __builtin_nontemporal_load(q); return *q; }
template<bool NT> __device__ inline void st(double* p,long long i,d2 v){ d2* q=reinterpret_cast<d2*>(p)+i; if(NT) __builtin_nontemporal_store(v,q); else *q=v; }

// FIN: 0 = release/acquire fences, 1 = sc1 stores + ticket + sc1 loads, 2 = partial rows only (second kernel), 3 = none
template<int FIN> __device__ inline void finalize(double (&acc)[NS], const P& p){
  __shared__ double sm[4][NS]; __shared__ int s_last;
  const int tid=threadIdx.x, lane=tid&63, wave=tid>>6;
#pragma unroll
  for(int s=0;s<NS;++s){ double v=wave_sum(acc[s]); if(lane==0) sm[wave][s]=v; }
  __syncthreads();
  if(FIN==3) { if(tid==0 && sm[0][0]==123.456) p.out[0]=1; return; }
  if(tid==0){
    double* row=p.partials+(size_t)blockIdx.x*NS;
    if(FIN==1){
#pragma unroll
      for(int s=0;s<NS;++s) __hip_atomic_store(row+s,(sm[0][s]+sm[1][s])+(sm[2][s]+sm[3][s]),__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned t=__hip_atomic_fetch_add(p.ticket,1u,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
      s_last=(t==gridDim.x-1);
    } else if(FIN==0){
#pragma unroll
      for(int s=0;s<NS;++s) row[s]=(sm[0][s]+sm[1][s])+(sm[2][s]+sm[3][s]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE,"agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned t=__hip_atomic_fetch_add(p.ticket,1u,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
      s_last=(t==gridDim.x-1);
    } else {
#pragma unroll
      for(int s=0;s<NS;++s) row[s]=(sm[0][s]+sm[1][s])+(sm[2][s]+sm[3][s]);
      s_last=0;
    }
  }
  if(FIN==2) return;
  __syncthreads();
  if(!s_last) return;
  if(FIN==0){ if(tid==0){ __builtin_amdgcn_fence(__ATOMIC_ACQUIRE,"agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } __syncthreads(); }
  double tot[NS];
#pragma unroll
  for(int s=0;s<NS;++s) tot[s]=0;
  for(unsigned b=tid;b<gridDim.x;b+=BLOCK){ const double* row=p.partials+(size_t)b*NS;
#pragma unroll
    for(int s=0;s<NS;++s) tot[s]+=__hip_atomic_load(row+s,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT); }
  __syncthreads();
#pragma unroll
  for(int s=0;s<NS;++s){ double v=wave_sum(tot[s]); if(lane==0) sm[wave][s]=v; }
  __syncthreads();
  if(tid==0){
#pragma unroll
    for(int s=0;s<NS;++s) p.out[s]=(sm[0][s]+sm[1][s])+(sm[2][s]+sm[3][s]);
    *p.ticket=0u; }
}

__global__ __launch_bounds__(BLOCK) void k_final2(P p,int nblocks){
  __shared__ double sm[4][NS];
  const int tid=threadIdx.x, lane=tid&63, wave=tid>>6;
  double tot[NS];
#pragma unroll
  for(int s=0;s<NS;++s) tot[s]=0;
  for(int b=tid;b<nblocks;b+=BLOCK){ const double* row=p.partials+(size_t)b*NS;
#pragma unroll
    for(int s=0;s<NS;++s) tot[s]+=row[s]; }
#pragma unroll
  for(int s=0;s<NS;++s){ double v=wave_sum(tot[s]); if(lane==0) sm[wave][s]=v; }
  __syncthreads();
  if(tid==0){
#pragma unroll
    for(int s=0;s<NS;++s) p.out[s]=(sm[0][s]+sm[1][s])+(sm[2][s]+sm[3][s]); }
}

struct L { d2 x,u,g,p; };
// KIND 0: accept_dir_trial (R x,u,g,p; W x,u,gt)  KIND 1: dir (R g,u; W u)  KIND 2: trial (R x,u,g,p; W gt)  KIND 3: copy (R x; W gt)
template<int KIND,bool NT> __device__ inline void load(const P& p,long long i,L& v){
  if(KIND==0||KIND==2||KIND==3) v.x=ld<NT>(p.x,i);
  if(KIND!=3) v.u=ld<NT>(p.u,i);
  if(KIND!=3) v.g=ld<NT>(p.g,i);
  if(KIND==0||KIND==2) v.p=ld<NT>(p.p0,i);
}
template<int KIND,bool NT> __device__ inline void body(const P& p,long long i,L& v,double (&acc)[NS]){
  if(KIND==3){ st<NT>(p.gt,i,v.x); return; }
  if(KIND==0){ v.x.x=v.x.x+p.a_acc*v.u.x; v.x.y=v.x.y+p.a_acc*v.u.y; st<NT>(p.x,i,v.x); }
  if(KIND==0||KIND==1){ d2 un; un.x=-v.g.x+p.beta*v.u.x; un.y=-v.g.y+p.beta*v.u.y;
    acc[7]+=v.g.x*un.x; acc[7]+=v.g.y*un.y; acc[8]+=un.x*un.x; acc[8]+=un.y*un.y; st<NT>(p.u,i,un); v.u=un; }
  if(KIND==0||KIND==2){ d2 xp,gt; xp.x=v.x.x+p.a_trial*v.u.x; xp.y=v.x.y+p.a_trial*v.u.y;
    gt.x=v.p.x*xp.x; gt.y=v.p.y*xp.y; acc[0]+=0.5*(gt.x*xp.x); acc[0]+=0.5*(gt.y*xp.y);
    st<NT>(p.gt,i,gt);
    acc[1]+=gt.x*v.u.x; acc[1]+=gt.y*v.u.y; acc[2]+=gt.x*gt.x; acc[2]+=gt.y*gt.y;
    double y0=gt.x-v.g.x,y1=gt.y-v.g.y;
    acc[3]+=gt.x*v.g.x; acc[3]+=gt.y*v.g.y; acc[4]+=y0*y0; acc[4]+=y1*y1; acc[5]+=v.u.x*y0; acc[5]+=v.u.y*y1; acc[6]+=y0*gt.x; acc[6]+=y1*gt.y; }
}

template<int KIND,int UNROLL,bool NT,int FIN>
__global__ __launch_bounds__(BLOCK) void k(P p){
  double acc[NS];
#pragma unroll
  for(int s=0;s<NS;++s) acc[s]=0;
  const long long n2=p.n>>1, T=(long long)gridDim.x*BLOCK;
  long long i=(long long)blockIdx.x*BLOCK+threadIdx.x;
  for(; i+(UNROLL-1)*T<n2; i+=UNROLL*T){
    L v[UNROLL];
#pragma unroll
    for(int k2=0;k2<UNROLL;++k2) load<KIND,NT>(p,i+k2*T,v[k2]);
#pragma unroll
    for(int k2=0;k2<UNROLL;++k2) body<KIND,NT>(p,i+k2*T,v[k2],acc);
  }
  for(; i<n2; i+=T){ L v; load<KIND,NT>(p,i,v); body<KIND,NT>(p,i,v,acc); }
  finalize<FIN>(acc,p);
}

// block-contiguous variant: each block owns a contiguous chunk (better DRAM page locality?)
template<int KIND,int UNROLL,bool NT,int FIN>
__global__ __launch_bounds__(BLOCK) void kc(P p){
  double acc[NS];
#pragma unroll
  for(int s=0;s<NS;++s) acc[s]=0;
  const long long n2=p.n>>1;
  const long long per=(n2+gridDim.x-1)/gridDim.x;
  const long long lo=per*blockIdx.x; long long hi=lo+per; if(hi>n2) hi=n2;
  long long i=lo+threadIdx.x;
  for(; i+(UNROLL-1)*BLOCK<hi; i+=UNROLL*BLOCK){
    L v[UNROLL];
#pragma unroll
    for(int k2=0;k2<UNROLL;++k2) load<KIND,NT>(p,i+k2*BLOCK,v[k2]);
#pragma unroll
    for(int k2=0;k2<UNROLL;++k2) body<KIND,NT>(p,i+k2*BLOCK,v[k2],acc);
  }
  for(; i<hi; i+=BLOCK){ L v; load<KIND,NT>(p,i,v); body<KIND,NT>(p,i,v,acc); }
  finalize<FIN>(acc,p);
}

__global__ void fill(double* v,long long n,double a,double b){ long long T=(long long)gridDim.x*blockDim.x; for(long long i=(long long)blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=T) v[i]=a+b*(double)(i%1000)/1000.0; }

static double bytes_of(int kind,long long n){ int v= kind==0?7: kind==1?3: kind==2?5:2; return 8.0*n*v; }

template<int KIND,int UNROLL,bool NT,int FIN,bool CONTIG>
void run(const char* name,P p,int grid,int reps,hipStream_t st,hipEvent_t e0,hipEvent_t e1){
  auto launch=[&](){ if(CONTIG) kc<KIND,UNROLL,NT,FIN><<<grid,BLOCK,0,st>>>(p); else k<KIND,UNROLL,NT,FIN><<<grid,BLOCK,0,st>>>(p); if(FIN==2) k_final2<<<1,BLOCK,0,st>>>(p,grid); };
  for(int w=0;w<3;++w) launch();
  CK(hipEventRecord(e0,st)); for(int r=0;r<reps;++r) launch(); CK(hipEventRecord(e1,st)); CK(hipStreamSynchronize(st));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=reps;
  double gb=bytes_of(KIND,p.n)/ms/1e6;
  printf("n=%.0e kind=%d %-28s grid=%5d  %9.1f us  %7.1f GB/s  %5.1f%%\n",(double)p.n,KIND,name,grid,ms*1e3,gb,gb/80.0);
}

int main(int argc,char** argv){
  hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long long nmax=100000000LL;
  const size_t pad=1<<20; // 1 MiB slack per array for staggering
  double *x,*u,*g,*gt,*p0,*partials,*out; unsigned* ticket;
  CK(hipMalloc(&x,nmax*8+pad)); CK(hipMalloc(&u,nmax*8+pad)); CK(hipMalloc(&g,nmax*8+pad)); CK(hipMalloc(&gt,nmax*8+pad)); CK(hipMalloc(&p0,nmax*8+pad));
  CK(hipMalloc(&partials,65536*NS*8)); CK(hipMalloc(&out,NS*8)); CK(hipMalloc(&ticket,4)); CK(hipMemset(ticket,0,4));
  fill<<<2048,256,0,st>>>(x,nmax+pad/8,1.0,0.1); fill<<<2048,256,0,st>>>(u,nmax+pad/8,-1.0,0.3); fill<<<2048,256,0,st>>>(g,nmax+pad/8,0.5,0.2); fill<<<2048,256,0,st>>>(p0,nmax+pad/8,1.0,9.0);
  CK(hipStreamSynchronize(st));
  printf("base addrs x=%p u=%p g=%p gt=%p p0=%p\n",x,u,g,gt,p0);
  long long ns[2]={10000000LL,100000000LL};
  for(int ni=0;ni<2;++ni){
    long long n=ns[ni]; int reps= n>=100000000LL?10:30;
    auto G=[&](int unroll,int cap){ long long n2=n/2; long long b=(n2+(long long)BLOCK*unroll-1)/((long long)BLOCK*unroll); if(b<1)b=1; if(b>cap)b=cap; return (int)b; };
    for(int stag=0;stag<4;++stag){
      size_t so = stag==0?0: stag==1?256/8: stag==2?4096/8: (65536+256)/8; // doubles
      P p{x,u+so,g+2*so,gt+3*so,p0+4*so,n,1e-9,0.5,1e-3,partials,ticket,out};
      printf("-- stagger %zu bytes\n",so*8);
      run<0,2,false,2,false>("U2 2k g2048",p,G(2,2048),reps,st,e0,e1);
      run<0,2,false,2,false>("U2 2k g1024",p,G(2,1024),reps,st,e0,e1);
      run<0,2,true,2,false>("U2 2k NT g2048",p,G(2,2048),reps,st,e0,e1);
      run<0,2,false,2,true>("U2 2k contig g2048",p,G(2,2048),reps,st,e0,e1);
      run<0,2,false,2,true>("U2 2k contig g4096",p,G(2,4096),reps,st,e0,e1);
      run<0,2,false,2,true>("U2 2k contig g8192",p,G(2,8192),reps,st,e0,e1);
      run<0,2,false,2,true>("U2 2k contig g16384",p,G(2,16384),reps,st,e0,e1);
      run<0,2,true,2,true>("U2 2k contig NT g4096",p,G(2,4096),reps,st,e0,e1);
      run<0,4,false,2,true>("U4 2k contig g4096",p,G(4,4096),reps,st,e0,e1);
      run<0,1,false,2,true>("U1 2k contig g4096",p,G(1,4096),reps,st,e0,e1);
      run<1,2,false,2,false>("dir U2 2k g2048",p,G(2,2048),reps,st,e0,e1);
      run<1,2,true,2,false>("dir U2 2k NT g2048",p,G(2,2048),reps,st,e0,e1);
      run<1,2,false,2,true>("dir U2 2k contig g4096",p,G(2,4096),reps,st,e0,e1);
      run<1,2,true,2,true>("dir U2 2k contig NT g4096",p,G(2,4096),reps,st,e0,e1);
      run<1,2,false,2,true>("dir U2 2k contig g8192",p,G(2,8192),reps,st,e0,e1);
      run<2,2,false,2,false>("trial U2 2k g2048",p,G(2,2048),reps,st,e0,e1);
      run<2,2,true,2,false>("trial U2 2k NT g2048",p,G(2,2048),reps,st,e0,e1);
      run<2,2,false,2,true>("trial U2 2k contig g4096",p,G(2,4096),reps,st,e0,e1);
      run<2,2,true,2,true>("trial U2 2k contig NT g4096",p,G(2,4096),reps,st,e0,e1);
      run<2,2,false,2,true>("trial U2 2k contig g8192",p,G(2,8192),reps,st,e0,e1);
      run<3,2,false,3,false>("copy U2 g2048",p,G(2,2048),reps,st,e0,e1);
      run<3,2,false,3,true>("copy U2 contig g4096",p,G(2,4096),reps,st,e0,e1);
      run<3,2,true,3,true>("copy U2 contig NT g4096",p,G(2,4096),reps,st,e0,e1);
    }
  }
  return 0;
}
