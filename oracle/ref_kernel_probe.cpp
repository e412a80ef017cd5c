// ref_kernel_probe — validation driver (TEST INFRASTRUCTURE ONLY): includes the reference's
// kernels.hpp / methods/jacobi.hpp from where they lie (-I$(REF)), calls each L1 kernel on
// seeded random data of many lengths and compares bit-for-bit with the oracle restatement.
// Built into oracle/_ref/ by `make probe`; run with OMP_NUM_THREADS=1.
#include "kernels.hpp"
#include "methods/jacobi.hpp"
#include "its_oracle.h"
#include <cstdint>
#include <cstring>
#include <vector>
static uint64_t sm(uint64_t &s){ uint64_t z=(s+=0x9E3779B97F4A7C15ULL); z=(z^(z>>30))*0xBF58476D1CE4E5B9ULL; z=(z^(z>>27))*0x94D049BB133111EBULL; return z^(z>>31);}
static double u(uint64_t &s){ return (double)(sm(s)>>11)/9007199254740992.0*2.0-1.0; }
static int same(const double*a,const double*b,size_t n){ return memcmp(a,b,8*n)==0; }
static int fails=0;
#define CHECK(name, cond) do{ if(!(cond)){ ++fails; printf("DIFF %-10s n=%d\n", name, n);} }while(0)
int main(){
    uint64_t s=42;
    std::vector<int> sizes; for(int n=1;n<=70;++n) sizes.push_back(n);
    for(int n : {127,128,129,255,256,1000,4095,4096,5001,65536}) sizes.push_back(n);
    for(int n : sizes){
        int per=27;
        std::vector<int> rp(n+1), col; std::vector<double> val, x(n), y1(n), y2(n), a(n), b(n), o1(n), o2(n), D(n);
        rp[0]=0; for(int i=0;i<n;++i){ int l=sm(s)%per; for(int k=0;k<l;++k){ col.push_back(sm(s)%n); val.push_back(u(s)); } rp[i+1]=col.size(); }
        for(int i=0;i<n;++i){ x[i]=u(s); a[i]=u(s); b[i]=u(s); D[i]=1.0+u(s)*0.5; }
        CRSMtxData<double> A; A.n_rows=n; A.n_cols=n; A.nnz=col.size(); A.row_ptr=rp.data(); A.col=col.data(); A.val=val.data();
        spmv_crs_cpu<double,double>(y1.data(), &A, x.data()); orc_spmv(n, rp.data(), col.data(), val.data(), x.data(), y2.data());
        CHECK("spmv", same(y1.data(),y2.data(),n));
        double d1; dot<double,double>(a.data(), b.data(), &d1, n); CHECK("dot", d1==orc_dot(a.data(), b.data(), n));
        dot<double,double>(a.data(), a.data(), &d1, n); CHECK("dot_self", d1==orc_dot(a.data(), a.data(), n));
        double sc=u(s);
        sum_vectors_cpu<double,double,double,double>(o1.data(), a.data(), b.data(), n, sc); orc_sum_vectors(o2.data(), a.data(), b.data(), n, sc);
        CHECK("sum", same(o1.data(),o2.data(),n));
        subtract_vectors_cpu<double,double,double,double>(o1.data(), a.data(), b.data(), n, sc); orc_subtract_vectors(o2.data(), a.data(), b.data(), n, sc);
        CHECK("subtract", same(o1.data(),o2.data(),n));
        o1=a; o2=a;   // aliased form out==a (gmres.hpp:314)
        subtract_vectors_cpu<double,double,double,double>(o1.data(), o1.data(), b.data(), n, sc); orc_subtract_vectors(o2.data(), o2.data(), b.data(), n, sc);
        CHECK("subtract_al", same(o1.data(),o2.data(),n));
        subtract_residual_cpu<double,double>(o1.data(), a.data(), b.data(), n); for(int i=0;i<n;++i) o2[i]=a[i]-b[i];
        CHECK("sub_resid", same(o1.data(),o2.data(),n));
        scale<double>(o1.data(), a.data(), sc, n); orc_scale(o2.data(), a.data(), sc, n);
        CHECK("scale", same(o1.data(),o2.data(),n));
        CHECK("nrm2", euclidean_vec_norm_cpu<double>(a.data(), n)==orc_nrm2(a.data(), n));
        CHECK("norm_inf", infty_vec_norm_cpu<double>(a.data(), n)==orc_norm_inf(a.data(), n));
        std::vector<int> lp(n+1), lc; std::vector<double> lv; lp[0]=0;
        for(int i=0;i<n;++i){ for(int k=rp[i];k<rp[i+1];++k) if(col[k]<i){ lc.push_back(col[k]); lv.push_back(val[k]*0.01);} lp[i+1]=lc.size(); }
        CRSMtxData<double> L; L.n_rows=n; L.n_cols=n; L.nnz=lc.size(); L.row_ptr=lp.data(); L.col=lc.data(); L.val=lv.data();
        std::fill(o1.begin(),o1.end(),0.0); std::fill(o2.begin(),o2.end(),0.0);
        spltsv_crs<double,double>(&L, o1.data(), D.data(), b.data()); orc_spltsv(n, lp.data(), lc.data(), lv.data(), o2.data(), D.data(), b.data());
        CHECK("spltsv", same(o1.data(),o2.data(),n));
        o1=a; o2=a;
        jacobi_normalize_x_cpu<double,double>(o1.data(), x.data(), D.data(), b.data(), n);
        for(int i=0;i<n;++i){ double adj=__builtin_fma(-D[i], x[i], o2[i]); o2[i]=(b[i]-adj)/D[i]; }
        CHECK("jac_norm", same(o1.data(),o2.data(),n));
    }
    printf("fails=%d\n", fails); return fails!=0;
}
