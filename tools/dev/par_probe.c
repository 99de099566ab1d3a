#include <omp.h>
#include <stdio.h>
#include <math.h>
int main(){ for (int T=1; T<=256; T*=2){ double t0=omp_get_wtime(); double tot=0;
#pragma omp parallel num_threads(T) reduction(+:tot)
{ double x=0; for (long i=0;i<200000000L/T;++i) x+=sin(i*1e-3); tot+=x; }
printf("T=%d %.3f s (%g)\n",T,omp_get_wtime()-t0,tot);} }
