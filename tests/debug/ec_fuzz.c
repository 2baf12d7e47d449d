/* Dev aid (CPU): the feeder with error concealment on randomly damaged streams (lost frames, cuts, bit flips) under the sanitizers:
 *   gcc -O1 -g -fsanitize=address,undefined -Iinclude -Ilibvpx.opencl_amd/csrc/host -o /tmp/ecfuzz tests/debug/ec_fuzz.c \
 *       libvpx.opencl_amd/csrc/host/vp8_parser.c libvpx.opencl_amd/csrc/host/vp8_tables.c -lpthread && /tmp/ecfuzz tests/golden/p_arf_176x144.ivf 7 40 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vp8_parser.h"
int main(int argc,char**argv){
    FILE*f=fopen(argv[1],"rb"); fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET); unsigned char*d=malloc(n); fread(d,1,n,f);
    unsigned seed=argc>2?atoi(argv[2]):1; int iters=argc>3?atoi(argv[3]):50;
    for(int it=0;it<iters;it++){
        srand(seed+it);
        vp8_parser*p=vp8_parser_create(); vp8_parser_set_error_concealment(p, 1);
        vp8ir_mb*mbs=NULL; int16_t*coef=NULL; vp8ir_mv*mvs=NULL; size_t cap=0;
        long off=32; int k=0, ok=0, err=0;
        while(off+12<=n){ unsigned sz; memcpy(&sz,d+off,4); const unsigned char*fr=d+off+12; off+=12+sz; k++;
            size_t use=sz; int mode=rand()%6;
            if(k>1){ if(mode==0) use=0; else if(mode==1) use=rand()%(sz+1); else if(mode==2) use=sz>40?sz-rand()%40:sz; }
            unsigned char*copy=malloc(use?use:1); memcpy(copy,fr,use);      /* exact-size heap copy: over-reads trip ASAN */
            if(mode==3 && use>8) copy[rand()%use]^=1<<(rand()%8);
            vp8ir_frame_hdr h;
            int rc=vp8_parser_begin_frame(p,use?copy:NULL,use,&h);
            if(!rc){ size_t nmb=(size_t)h.mb_cols*h.mb_rows; if(nmb>cap){cap=nmb; mbs=realloc(mbs,nmb*64); coef=realloc(coef,nmb*800); mvs=realloc(mvs,nmb*64);}
                int corrupt; rc=vp8_parser_decode_mbs(p,mbs,coef,mvs,&corrupt); }
            if(rc) err++; else ok++;
            free(copy);
        }
        vp8_parser_destroy(p); free(mbs); free(coef); free(mvs);
        if(it%10==0) printf("iter %d: %d ok %d errors\n",it,ok,err);
    }
    return 0;
}
