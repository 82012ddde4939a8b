!> Error convention and multi-dimensional use of the drop-in module: ierror codes of
!! splcw (:674-686) / splde (:1155-1161), splcc, a 2-D fit with derivative evaluation.
program test_api
    use splpak_module, wp => splpak_wp
    implicit none
    integer,parameter :: m = 2000
    integer :: nodes(2), ierror, i, nbad
    real(wp) :: xdata(2,m), ydata(m), xmin(2), xmax(2), coef(64), work(64*65), x(2), f, fx, u
    real(wp) :: xs(2,50), fs(50), jet(6,50)
    integer(8) :: s
    type(splpak_type) :: solver
    character(len=16) :: backend

    ! `<program> host`: the same scenario on the module's HOST solver (set_host; no GPU needed)
    call get_command_argument(1, backend)
    if (trim(backend) == 'host') call solver%set_host(.true.)

    nbad = 0
    s = 42_8
    do i = 1, m
        s = mod(48271_8*s, 2147483647_8); u = real(s,wp)/2147483647.0_wp; xdata(1,i) = u
        s = mod(48271_8*s, 2147483647_8); u = real(s,wp)/2147483647.0_wp; xdata(2,i) = u
        ydata(i) = 1.0_wp + 2.0_wp*xdata(1,i) - 3.0_wp*xdata(2,i)
    end do
    xmin = 0.0_wp; xmax = 1.0_wp; nodes = [8,8]

    ! error codes, first failing check wins
    call solver%initialize(0,xdata,2,ydata,m,xmin,xmax,nodes,0.0_wp,coef,64,work,64*65,ierror); call expect(ierror,101)
    call solver%initialize(2,xdata,2,ydata,m,xmin,xmax,[3,8],0.0_wp,coef,64,work,64*65,ierror); call expect(ierror,102)
    call solver%initialize(2,xdata,2,ydata,m,xmin,xmin,nodes,0.0_wp,coef,64,work,64*65,ierror); call expect(ierror,103)
    call solver%initialize(2,xdata,2,ydata,m,xmin,xmax,nodes,0.0_wp,coef,63,work,64*65,ierror); call expect(ierror,104)
    call solver%initialize(2,xdata,2,ydata,0,xmin,xmax,nodes,0.0_wp,coef,64,work,64*65,ierror); call expect(ierror,105)
    call solver%initialize(2,xdata,2,ydata,m,xmin,xmax,nodes,1.0_wp,coef,64,work,10,ierror);    call expect(ierror,106)
    call solver%initialize(2,xdata,2,ydata,5,xmin,xmax,nodes,0.0_wp,coef,64,work,64*65,ierror); call expect(ierror,107)

    ! splcc: plane is reproduced exactly without smoothing rows (natural splines contain linears)
    call solver%initialize(2,xdata,2,ydata,m,xmin,xmax,nodes,0.0_wp,coef,64,work,64*65,ierror); call expect(ierror,0)
    x = [0.3_wp, 0.7_wp]
    f = solver%evaluate(2,x,coef,xmin,xmax,nodes,ierror); call expect(ierror,0)
    if (abs(f - (1.0_wp + 2.0_wp*x(1) - 3.0_wp*x(2))) > 1.0e-10_wp) call fail('plane value')
    fx = solver%evaluate(2,x,[1,0],coef,xmin,xmax,nodes,ierror); call expect(ierror,0)
    if (abs(fx - 2.0_wp) > 1.0e-9_wp) call fail('d/dx1')
    fx = solver%evaluate(2,x,[0,1],coef,xmin,xmax,nodes,ierror); call expect(ierror,0)
    if (abs(fx + 3.0_wp) > 1.0e-9_wp) call fail('d/dx2')
    fx = solver%evaluate(2,[1.5_wp,-0.25_wp],coef,xmin,xmax,nodes,ierror)      ! linear extrapolation
    if (abs(fx - (1.0_wp + 3.0_wp + 0.75_wp)) > 1.0e-9_wp) call fail('extrapolation')
    f = solver%evaluate(2,x,[3,0],coef,xmin,xmax,nodes,ierror); call expect(ierror,104)
    f = solver%evaluate(2,x,coef,xmin,xmax,[3,8],ierror);       call expect(ierror,102)

    do i = 1, 50
        xs(:,i) = xdata(:,i)
    end do
    call solver%evaluate_many(2,50,xs,2,[0,1],coef,xmin,xmax,nodes,fs,ierror); call expect(ierror,0)
    if (maxval(abs(fs + 3.0_wp)) > 1.0e-9_wp) call fail('evaluate_many derivative')

    ! value + gradient + Hessian in one pass: plane -> gradient (2,-3), zero Hessian; columns = splde patterns
    call solver%evaluate_derivatives(2,50,xs,2,2,coef,xmin,xmax,nodes,jet,6,ierror); call expect(ierror,0)
    if (maxval(abs(jet(2,:) - 2.0_wp)) > 1.0e-9_wp .or. maxval(abs(jet(3,:) + 3.0_wp)) > 1.0e-9_wp) &
        call fail('evaluate_derivatives gradient')
    if (maxval(abs(jet(4:6,:))) > 1.0e-6_wp) call fail('evaluate_derivatives Hessian of a plane')
    do i = 1, 50, 7
        f = solver%evaluate(2,xs(:,i),coef,xmin,xmax,nodes,ierror)
        if (abs(jet(1,i) - f) > 1.0e-12_wp*max(1.0_wp,abs(f))) call fail('evaluate_derivatives value')
        fx = solver%evaluate(2,xs(:,i),[1,1],coef,xmin,xmax,nodes,ierror)
        if (abs(jet(5,i) - fx) > 1.0e-9_wp) call fail('evaluate_derivatives mixed derivative')
    end do
    call solver%evaluate_derivatives(2,50,xs,2,1,coef,xmin,xmin,nodes,jet,6,ierror); call expect(ierror,103)

    ! named options of the HIP library (round 6): an unknown name is refused, a known one is taken -- and the fit that follows,
    ! with the iterative solve in front of the factorisation, returns the same plane
    call solver%set_option('no_such_option','1',ierror); call expect(ierror,-3)
    call solver%set_option('solver','pcg+direct',ierror); call expect(ierror,0)
    coef = 0.0_wp
    call solver%initialize(2,xdata,2,ydata,m,xmin,xmax,nodes,0.0_wp,coef,64,work,64*65,ierror); call expect(ierror,0)
    f = solver%evaluate(2,x,coef,xmin,xmax,nodes,ierror)
    if (abs(f - (1.0_wp + 2.0_wp*x(1) - 3.0_wp*x(2))) > 1.0e-10_wp) call fail('plane value with solver = pcg+direct')
    call solver%set_option('solver','direct',ierror); call expect(ierror,0)

    if (nbad /= 0) error stop 'test_api FAILED'
    write(*,*) 'PASS test_api'
contains
    subroutine expect(got,want)
        integer,intent(in) :: got, want
        if (got /= want) then
            write(*,*) 'expected ierror', want, ' got', got
            nbad = nbad + 1
        end if
    end subroutine expect
    subroutine fail(what)
        character(len=*),intent(in) :: what
        write(*,*) 'FAILED: ', what
        nbad = nbad + 1
    end subroutine fail
end program test_api
