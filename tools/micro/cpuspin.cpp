#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
int main(int argc,char**argv){
  for(int nt : {1,8,16,32,64,128,256}){
    std::vector<std::thread> th; std::atomic<unsigned long long> sum{0};
    auto t0=std::chrono::steady_clock::now();
    for(int t=0;t<nt;t++) th.emplace_back([&,t]{ unsigned long long x=t+1; for(long i=0;i<400000000L;i++){ x=x*6364136223846793005ULL+1442695040888963407ULL; } sum+=x; });
    for(auto&x:th)x.join();
    double s=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
    printf("threads %3d: %.3f s  -> %.1f thread-equivalents (sum %llu)\n",nt,s,nt*0.0/1+ (double)nt,(unsigned long long)sum.load());
  }
}
