// dp_table_bench.cpp -- ONE thread inserting random points into the distinguished-point table (kng_dptable): ns per insert by
// table size, with the process's page faults and system time, to tell the table's own cost from the kernel's (memory growth).
// build: g++ -O2 -std=c++17 -Ikangaroo_amd/host -o tools/dp_table_bench tools/dp_table_bench.cpp -Lkangaroo_amd/lib -lkangaroo_host \
//            -lkangaroo_hip -Wl,-rpath,'$ORIGIN/../kangaroo_amd/lib' -lpthread
#include <sys/resource.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
#include "kng_dptable.h"
static inline uint64_t xs(uint64_t &s){ s^=s<<13; s^=s>>7; s^=s<<17; return s;}
int main(int argc,char**argv){
  size_t N = argc>1? atol(argv[1]) : 4000000;
  const size_t PD = argc>2? atol(argv[2]) : 4; // prefetch distance between stages
  kngt_table*t=kngt_create();
  uint64_t st=88172645463325252ULL;
  std::vector<kngt_entry> e(N); std::vector<uint32_t> b(N);
  for(size_t i=0;i<N;i++){ e[i].x[0]=xs(st); e[i].x[1]=xs(st); e[i].d[0]=xs(st); e[i].d[1]=xs(st)&0x3FFF; b[i]=xs(st)&(KNGT_BUCKETS-1);}
  for(int rep=0;rep<4;rep++){
  auto t0=std::chrono::steady_clock::now();
  size_t lo=rep*N/4, hi=(rep+1)*N/4;
  for(size_t i=lo;i<hi;i++){
    if(i+3*PD<hi) kngt_prefetch(t,b[i+3*PD],e[i+3*PD].x[1],0);
    if(i+2*PD<hi) kngt_prefetch(t,b[i+2*PD],e[i+2*PD].x[1],1);
    if(i+PD<hi) kngt_prefetch(t,b[i+PD],e[i+PD].x[1],2);
    kngt_entry o; kngt_add_entry(t,b[i],&e[i],&o);}
  double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
  struct rusage ru; getrusage(RUSAGE_SELF,&ru);
  printf("quarter %d: %.1f ns/insert, table %.2f GiB, %llu items; so far %ld minor faults, %.2f s user, %.2f s system\n",rep,dt/(hi-lo)*1e9,kngt_memory_bytes(t)/1073741824.0,(unsigned long long)kngt_count(t),
         ru.ru_minflt, ru.ru_utime.tv_sec+ru.ru_utime.tv_usec*1e-6, ru.ru_stime.tv_sec+ru.ru_stime.tv_usec*1e-6);}
}
